from .trainer import Trainer, get_trainer

__all__ = ["Trainer", "get_trainer"]
