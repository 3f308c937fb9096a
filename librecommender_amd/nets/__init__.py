"""Device-side computation graphs of the north-star models (what `build_model()` creates in the
reference), expressed over the HIP hot-path kernels + torch dense layers."""
from .fm_nets import DeepFMNet, FMNet, ShardedDeepFMNet

__all__ = ["DeepFMNet", "FMNet", "ShardedDeepFMNet"]
from .tower_nets import TwoTowerNet  # noqa: E402

__all__.append("TwoTowerNet")
from .feat_embedding import FeatEmbedding, FeatSpec  # noqa: E402
from .feat_nets import FeatDeepFMNet, FeatDINNet, FeatFMNet  # noqa: E402

__all__ += ["FeatEmbedding", "FeatSpec", "FeatDeepFMNet", "FeatDINNet", "FeatFMNet"]
from .field_parallel import FieldParallelDeepFMNet  # noqa: E402
from .ngcf_net import NGCFNet  # noqa: E402

__all__ += ["FieldParallelDeepFMNet", "NGCFNet"]
