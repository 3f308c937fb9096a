"""Epoch loop (`libreco/training/tf_trainer.py:46-101`, `torch_trainer.py:77-138`): seeded
`RandomSampler` order, host collator (negative sampling etc.), one `model.train_on_batch` per
batch.  The loss stays on the device; it is read back once per epoch."""
from __future__ import annotations

import torch

from ..batch import adjust_batch_size, get_batch_loader
from ..batch.device_loader import DevicePointwiseLoader
from ..layers.tail import check_all as tail_check
from ..nets.din_fused import lazy_join
from ..utils.misc import colorize, time_block


def _with_next(batches):
    """(batch, the batch after it or None) over an iterable: one batch of lookahead."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class Trainer:
    def __init__(self, model):
        self.model = model
        self.n_epochs = model.n_epochs
        self.batch_size = adjust_batch_size(model, model.batch_size)

    def run(self, train_data, neg_sampling, verbose, shuffle, eval_data, metrics, k, eval_batch_size,
            eval_user_num, num_workers):
        from ..evaluation import print_metrics

        m = self.model
        loader = get_batch_loader(m, train_data, neg_sampling, self.batch_size, shuffle, num_workers, m.seed)
        for epoch in range(1, self.n_epochs + 1):
            if getattr(m, "lr_decay", False) and verbose > 0:
                print(f"With lr_decay, epoch {epoch} learning rate: {m.current_lr()}")
            with time_block(f"Epoch {epoch}", verbose):
                # device-side loader + graph-replayed steps: the next batch is collated beside the running step
                with lazy_join(isinstance(loader, DevicePointwiseLoader), model=m):
                    if getattr(m, "takes_next_batch", lambda: False)():
                        # row-sharded tables: the NEXT batch's exchange plan (de-duplication + per-peer counts, the step's only
                        # host read) is built beside this step — the model is told which batch comes next (`next_idx`)
                        losses = [m.train_on_batch(b, next_batch=nb) for b, nb in _with_next(loader)]
                    else:
                        losses = [m.train_on_batch(b) for b in loader]
                m.on_epoch_end(epoch)
                tail_check()        # a one-launch tail that gave up on a grid barrier raises here, not never
            if verbose > 1:
                mean = float(torch.stack(losses).mean()) if losses else float("nan")
                print("\t " + colorize(f"train_loss: {round(mean, 4)}", "green"))
                m.prepare_for_eval()
                print_metrics(model=m, neg_sampling=neg_sampling, eval_data=eval_data, metrics=metrics,
                              eval_batch_size=eval_batch_size, k=k, sample_user_num=eval_user_num,
                              seed=m.seed)
                print("=" * 30)


def get_trainer(model):
    """`training/dispatch.py:6-42`: the reference picks a TF or torch trainer class by model name;
    here one epoch loop serves every model (the optimiser lives in the model's net)."""
    return Trainer(model)
