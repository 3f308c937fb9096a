from .cold_start import cold_start_rec, popular_recommendations
from .ranking import rank_recommendations
from .recommend import ConsumedIndex, construct_rec, check_dynamic_rec_feats, recommend_from_embedding

__all__ = ["cold_start_rec", "popular_recommendations", "rank_recommendations", "ConsumedIndex", "construct_rec",
           "check_dynamic_rec_feats", "recommend_from_embedding"]
