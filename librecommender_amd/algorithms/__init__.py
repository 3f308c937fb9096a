from .two_tower import TwoTower

__all__ = ["TwoTower"]
