"""Multi-GPU sharding of the plane-sweep path (one process per GPU, ``torch.distributed``: RCCL over xGMI
with backend "nccl" on MI355X, gloo in the CPU tests).

Two shardings (the reference has neither -- its only parallelism is data-parallel DDP):

* reference-view shard -- independent objects, no collective: ``view_shard`` (what ``bench.py --gpus N`` runs);
* depth-plane shard -- rank r owns planes [d0, d1): the fused warp+cost kernel needs no communication (feature
  maps are replicated, ~1.3 MB per view), and the softmax over D is merged from per-rank partials
  ``(max, sum e, sum e*depth, sum e*index)`` (``pscv_softargmin`` ``out_partials``) with ONE small all-gather
  (4*h*w floats per rank) followed by a local log-sum-exp merge: ``merge_partials``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plane_shard(D: int, world: int, rank: int, multiple: int = 8) -> Tuple[int, int]:
    """Contiguous plane range of ``rank``; boundaries are multiples of ``multiple`` (8 for MVSNet's U-Net,
    2 for the Vis U-Net) and the remainder goes to the last ranks."""
    if D % multiple:
        raise ValueError(f"D={D} is not a multiple of {multiple}")
    units = D // multiple
    base, extra = divmod(units, world)
    counts = [base + (1 if r >= world - extra else 0) for r in range(world)]
    d0 = sum(counts[:rank]) * multiple
    return d0, d0 + counts[rank] * multiple


def view_shard(n_items: int, world: int, rank: int) -> List[int]:
    """Indices of the reference views (batch items) rank ``rank`` processes: round-robin, no collective."""
    return list(range(rank, n_items, world))


def merge_partials_local(parts: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Log-sum-exp merge of per-shard partials, each [B,4,h,w] = (max, sum e, sum e*depth, sum e*index)
    -> (depth [B,h,w], expected index [B,h,w])."""
    stack = torch.stack(list(parts))                       # [S,B,4,h,w]
    m = stack[:, :, 0].max(dim=0).values                   # [B,h,w]
    scale = torch.exp(stack[:, :, 0] - m.unsqueeze(0))     # [S,B,h,w]
    se = (stack[:, :, 1] * scale).sum(0)
    sd = (stack[:, :, 2] * scale).sum(0)
    si = (stack[:, :, 3] * scale).sum(0)
    return sd / se, si / se


def merge_partials(partials: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
    """All ranks contribute their shard's partials [B,4,h,w]; everyone gets the merged (depth, index).
    One all-gather of 4*B*h*w floats per rank (330 KB at 128x160): latency-bound, far below the per-link
    xGMI bandwidth, so a direct all-gather is the right collective (no ring needed)."""
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(partials) for _ in range(world)]
    dist.all_gather(bufs, partials.contiguous(), group=group)
    return merge_partials_local(bufs)


class CollectiveTrace:
    """Measurement aid (bench.py ``sharded`` legs): while active, every ``torch.distributed`` collective the sharded models issue
    (``all_reduce``, ``all_gather``, ``broadcast``) is bracketed by CUDA events on the current stream and recorded as
    ``(name, payload bytes of this rank, milliseconds)``.  ``summary()`` groups the records by (name, bytes).  The events add a
    few microseconds per call; the traced pass is therefore separate from the timed one."""

    NAMES = ("all_reduce", "all_gather", "broadcast", "reduce_scatter_tensor", "all_gather_into_tensor")

    def __init__(self):
        self.records = []
        self._saved = {}

    def _wrap(self, name, fn):
        def traced(*args, **kwargs):
            tensors = [a for a in args if torch.is_tensor(a)] + [x for a in args if isinstance(a, (list, tuple)) for x in a if torch.is_tensor(x)]
            payload = max((x.numel() * x.element_size() for x in tensors), default=0)
            if not tensors or not tensors[0].is_cuda:
                return fn(*args, **kwargs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kwargs)
            e1.record()
            self.records.append((name, payload, e0, e1))
            return out
        return traced

    def __enter__(self):
        for name in self.NAMES:
            if hasattr(dist, name):
                self._saved[name] = getattr(dist, name)
                setattr(dist, name, self._wrap(name, self._saved[name]))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        return False

    def summary(self):
        torch.cuda.synchronize()
        groups = {}
        for name, payload, e0, e1 in self.records:
            g = groups.setdefault((name, payload), [0, 0.0])
            g[0] += 1
            g[1] += e0.elapsed_time(e1)
        return [{"collective": k[0], "bytes_per_rank": k[1], "calls": v[0], "avg_us": v[1] / v[0] * 1e3} for k, v in sorted(groups.items())]
