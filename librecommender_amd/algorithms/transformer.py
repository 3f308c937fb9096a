"""`Transformer` (`libreco/algorithms/transformer.py`): same constructor and errors; embedding rows and their Adam
update on the HIP kernels, the small [B, L, D] attention blocks as device torch ops (`nets/seq_nets.py`)."""
from __future__ import annotations

from ..bases.base import hip_device
from ..nets import FeatSpec
from ..nets.seq_nets import FeatTransformerNet
from .din import DIN


class Transformer(DIN):
    """Sequence handling, `recommend_user(seq=...)` and full-catalogue ranking are DIN's."""

    def __init__(self, task, data_info=None, loss_type="cross_entropy", embed_size=16, n_epochs=1, lr=0.001,
                 lr_decay=False, epsilon=1e-5, reg=None, batch_size=256, sampler="random", num_neg=1, use_bn=True,
                 dropout_rate=None, hidden_units=(128, 64, 32), recent_num=10, random_num=None, num_heads=1,
                 num_tfm_layers=1, positional_embedding="trainable", use_causal_mask=False, feat_agg_mode="concat",
                 multi_sparse_combiner="sqrtn", seed=42, lower_upper_bound=None, tf_sess_config=None, device="cuda",
                 dense_adam=False):
        super().__init__(task, data_info, loss_type, embed_size, n_epochs, lr, lr_decay, epsilon, reg, batch_size,
                         sampler, num_neg, use_bn, dropout_rate, hidden_units, recent_num, random_num, False,
                         multi_sparse_combiner, seed, lower_upper_bound, tf_sess_config, device, dense_adam, False)
        self.all_args = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        self.num_heads, self.num_tfm_layers = num_heads, num_tfm_layers
        self.positional_embedding, self.use_causal_mask, self.feat_agg_mode = positional_embedding, use_causal_mask, feat_agg_mode
        if self.task == "ranking" and self.loss_type not in ("cross_entropy", "focal"):
            raise ValueError(f"unsupported `loss_type`: {self.loss_type}")
        if self.feat_agg_mode not in ("concat", "elementwise"):
            raise ValueError("`feat_agg_mode` must be `concat` or `elementwise`.")

    def build_model(self):
        self.device = hip_device(self._device_arg)
        d = self.data_info
        self.net = FeatTransformerNet(FeatSpec.from_data_info(d, self.multi_sparse_combiner), self.embed_size,
                                      self.hidden_units, self.use_bn, self.dropout_rate, self.max_seq_len,
                                      self.num_heads, self.num_tfm_layers, self.positional_embedding,
                                      self.use_causal_mask, self.feat_agg_mode, d.item_sparse_unique,
                                      d.item_dense_unique, d.item_dense_col.index, self.lr, self.epsilon, self.seed,
                                      self.device, self.dense_adam, self.reg)
