"""Loss bookkeeping base class of the training harness -- the counterpart of the reference's ``utils/trainer.py`` (47 lines:
running sums per loss name, an all-reduced epoch mean, a per-``print_every`` running mean).  The harness is a caller of the
plane-sweep path; it is mirrored so that ``models.trainer.Trainer`` carries every method the reference's
``train.py`` calls (``step``, ``test``, ``log_iter``, ``log_epoch``, ``keep_losses``) when this package is installed as ``models``."""
from __future__ import annotations

import torch.distributed as dist


class Trainer:
    def __init__(self):
        self.loss_means = {}            # name -> sum since the last log_epoch()
        self.loss_running_means = {}    # name -> sum since the last log_iter() (training losses only)
        self.nb_iter = 0
        self.ims = {}

    def keep_losses(self, losses):
        """Accumulates detached loss tensors; names starting with "train" also feed the running mean (utils/trainer.py:40-46)."""
        for name, value in losses.items():
            if name.startswith("train"):
                self.loss_running_means[name] = self.loss_running_means.get(name, 0) + value
            self.loss_means[name] = self.loss_means.get(name, 0) + value

    def log_iter(self):
        """Mean over the last ``args.print_every`` iterations, then reset (utils/trainer.py:35-38)."""
        out = {name: total / self.args.print_every for name, total in self.loss_running_means.items()}
        self.loss_running_means = {}
        return out

    def log_epoch(self, epoch):
        """Mean over the iterations since the last call, averaged over the ranks of the default process group, then reset
        (utils/trainer.py:23-33)."""
        out = {name: total / self.nb_iter for name, total in self.loss_means.items()}
        self.loss_means, self.nb_iter = {}, 0
        world = dist.get_world_size()
        for name in out:
            dist.all_reduce(tensor=out[name], op=dist.ReduceOp.SUM)
            out[name] /= world
        out["epoch"] = epoch
        return out
