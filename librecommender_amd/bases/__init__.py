from .base import Base
from .embed_base import EmbedBase
from .feat_base import FeatBase

__all__ = ["Base", "EmbedBase", "FeatBase"]
