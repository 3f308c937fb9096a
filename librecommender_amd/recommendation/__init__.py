from .cold_start import cold_start_rec
from .recommend import ConsumedIndex, construct_rec, check_dynamic_rec_feats, recommend_from_embedding

__all__ = ["cold_start_rec", "ConsumedIndex", "construct_rec", "check_dynamic_rec_feats", "recommend_from_embedding"]
