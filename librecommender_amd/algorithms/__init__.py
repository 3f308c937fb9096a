from .lightgcn import LightGCN
from .two_tower import TwoTower

__all__ = ["LightGCN", "TwoTower"]
