from .export import save_embed, save_online

__all__ = ["save_embed", "save_online"]
