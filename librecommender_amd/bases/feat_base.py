"""Placeholder import target; the feature-model base (FM / DeepFM / DIN) lives in feat_base2 once
built."""
from .base import Base


class FeatBase(Base):
    pass
