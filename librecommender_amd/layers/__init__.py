from .dense import DenseStack, TFBatchNorm, TFDense, DenseParams
from .embedding import FieldTables

__all__ = ["DenseStack", "TFBatchNorm", "TFDense", "DenseParams", "FieldTables"]
