from .din import DIN
from .fm import FM, DeepFM
from .lightgcn import LightGCN
from .ngcf import NGCF
from .sim import SIM
from .transformer import Transformer
from .two_tower import TwoTower
from .youtube_ranking import YouTubeRanking
from .youtube_retrieval import YouTubeRetrieval

__all__ = ["DIN", "DeepFM", "FM", "LightGCN", "NGCF", "SIM", "Transformer", "TwoTower", "YouTubeRanking", "YouTubeRetrieval"]
