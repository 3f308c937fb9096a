from .din import DIN
from .fm import FM, DeepFM
from .lightgcn import LightGCN
from .two_tower import TwoTower

__all__ = ["DIN", "DeepFM", "FM", "LightGCN", "TwoTower"]
