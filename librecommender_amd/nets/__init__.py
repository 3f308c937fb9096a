"""Device-side computation graphs of the north-star models (what `build_model()` creates in the
reference), expressed over the HIP hot-path kernels + torch dense layers."""
from .feat_embedding import FeatEmbedding, FeatSpec
from .feat_nets import FeatDeepFMNet, FeatDINNet, FeatFMNet, FeatYouTubeRankingNet, ShardedDINNet
from .field_parallel import FieldParallelDeepFMNet
from .fm_nets import DeepFMNet, FMNet, ShardedDeepFMNet, ShardedFMNet
from .ngcf_net import NGCFNet
from .tower_nets import ShardedTwoTowerNet, TwoTowerNet

__all__ = ["DeepFMNet", "FMNet", "ShardedDeepFMNet", "ShardedFMNet", "FieldParallelDeepFMNet", "TwoTowerNet", "ShardedTwoTowerNet", "FeatEmbedding",
           "FeatSpec", "FeatDeepFMNet", "FeatDINNet", "FeatFMNet", "FeatYouTubeRankingNet", "ShardedDINNet", "NGCFNet"]
