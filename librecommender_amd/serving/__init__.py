from .embed_server import EmbedServer, InvalidUser
from .export import save_embed, save_online

__all__ = ["EmbedServer", "InvalidUser", "save_embed", "save_online"]
