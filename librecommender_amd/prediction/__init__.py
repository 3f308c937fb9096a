from .predict import convert_id, normalize_prediction, predict_from_embedding

__all__ = ["convert_id", "normalize_prediction", "predict_from_embedding"]
