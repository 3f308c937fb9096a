"""`YouTubeRanking` (`libreco/algorithms/youtube_ranking.py`): same constructor, HIP path — user / item / feature
embeddings plus the sqrt(len)-scaled sum of the user's recent item embeddings (`lr_embed_bag_pool_f32`) into an MLP."""
from __future__ import annotations

from ..bases.base import hip_device
from ..nets import FeatSpec, FeatYouTubeRankingNet
from .din import DIN


class YouTubeRanking(DIN):
    """Sequence handling (recent / random windows, `recommend_user(seq=...)`, device-side collation) is DIN's."""

    def __init__(self, task="ranking", data_info=None, loss_type="cross_entropy", embed_size=16, n_epochs=20,
                 lr=0.001, lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random",
                 num_neg=1, use_bn=True, dropout_rate=None, hidden_units=(128, 64, 32), recent_num=10,
                 random_num=None, multi_sparse_combiner="sqrtn", seed=42, lower_upper_bound=None,
                 tf_sess_config=None, device="cuda", dense_adam=False, device_sampling=False):
        assert task == "ranking", "YouTube models is only suitable for ranking"
        super().__init__(task, data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, reg, batch_size,
                         sampler, num_neg, use_bn, dropout_rate, hidden_units, recent_num, random_num, False,
                         multi_sparse_combiner, seed, lower_upper_bound, tf_sess_config, device, dense_adam,
                         device_sampling)
        self.all_args = {k: v for k, v in locals().items() if k not in ("self", "__class__")}

    def build_model(self):
        self.device = hip_device(self._device_arg)
        self.net = FeatYouTubeRankingNet(FeatSpec.from_data_info(self.data_info, self.multi_sparse_combiner),
                                         self.embed_size, self.hidden_units, self.use_bn, self.dropout_rate,
                                         self.max_seq_len, self.lr, self.epsilon, self.seed, self.device,
                                         self.dense_adam, self.reg)
