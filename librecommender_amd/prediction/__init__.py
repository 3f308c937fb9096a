from .predict import convert_id, normalize_prediction, predict_data_with_feats, predict_from_embedding
from .preprocess import catalog_features, features_from_batch, user_tower_features

__all__ = ["convert_id", "normalize_prediction", "predict_from_embedding", "predict_data_with_feats",
           "features_from_batch", "catalog_features", "user_tower_features"]
