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


def merge_partials_stats(partials: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
    """Like ``merge_partials`` but also returns the merged softmax statistics: (depth, index, max, sum exp), each [B,h,w]."""
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(partials) for _ in range(world)]
    dist.all_gather(bufs, partials.contiguous(), group=group)
    stack = torch.stack(bufs)
    m = stack[:, :, 0].max(dim=0).values
    scale = torch.exp(stack[:, :, 0] - m.unsqueeze(0))
    se = (stack[:, :, 1] * scale).sum(0)
    return (stack[:, :, 2] * scale).sum(0) / se, (stack[:, :, 3] * scale).sum(0) / se, m, se


def photometric_confidence_shard(own_logits: torch.Tensor, m: torch.Tensor, Z: torch.Tensor, index: torch.Tensor, a: int,
                                 group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """MVSNet / CVP photometric confidence (reference models/MVSNet/model.py:211-215) with the depth planes sharded: this rank's
    logits [B,n,h,w] are planes [a, a+n); under the merged statistics (max m, sum of exp Z, expected index) every rank adds the
    probabilities of the planes i-1 .. i+2 it owns, i = trunc(E[index]); one all-reduce of a [B,h,w] map finishes the sum."""
    n = own_logits.shape[1]
    i0 = index.to(torch.int64)                                        # trunc: the expected index is >= 0
    conf = torch.zeros_like(m)
    for k in (-1, 0, 1, 2):
        g = i0 + k - a
        ok = (g >= 0) & (g < n)
        val = torch.gather(own_logits, 1, g.clamp(0, n - 1).unsqueeze(1)).squeeze(1)
        conf = conf + torch.where(ok, torch.exp(val - m) / Z, torch.zeros_like(m))
    dist.all_reduce(conf, group=group)
    return conf


def halo_sync(y_ext: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """Depth-plane shard of a 3-tap-per-layer U-Net: ``y_ext`` [1, n + 4, ...] holds a layer's output on this rank's n owned
    planes at [2, n + 2) with two halo slots per side.  The slot next to the owned planes is filled with the neighbour's boundary
    plane (one point-to-point message per neighbour: C x h x w elements, 328 KB for MVSNet's first layer at 128 x 160); at the
    ends of the volume the halo slots are zeroed = the convolution's own padding.  The outer slots only ever feed outputs that
    are discarded."""
    if y_ext.shape[0] != 1:
        raise ValueError("halo_sync: one batch item at a time (plane slices must be contiguous)")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = y_ext.shape[1] - 4
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    sends, recvs = [], []
    if rank > 0:
        sends.append((y_ext[:, 2], peer(rank - 1)))
        recvs.append((y_ext[:, 1], peer(rank - 1)))
    else:
        y_ext[:, :2].zero_()
    if rank + 1 < world:
        sends.append((y_ext[:, n + 1], peer(rank + 1)))
        recvs.append((y_ext[:, n + 2], peer(rank + 1)))
    else:
        y_ext[:, n + 2:].zero_()
    exchange(sends, recvs, group)


# ---- slab exchange of the Vis-MVSNet source-view shard (reduce-scatter -> slab-sharded RegFuse) ----------
FUSE_HALO = 8      # reach of the fuse U-Net (+-7) plus its 3x3x3 head (+-1) along every axis, in voxels (scripts/dev/depth_shard_probe.py)


def slab_size(extent: int, world: int) -> int:
    """Units per rank along a sharded axis: even (the U-Net's stride-2 phase) and equal on every rank (reduce_scatter_tensor needs
    equal chunks; the tail rank(s) own fewer VALID units when world * size > extent)."""
    return 2 * ((extent + 2 * world - 1) // (2 * world))


def slab_axis(d: int, h: int, world: int, halo: int = FUSE_HALO):
    """Which axis of a [n,d,h,w,c] volume to cut into per-rank slabs: depth (1) or rows (2), whichever leaves the thicker slab
    (least halo recompute); None when neither slab is at least ``halo`` thick (then a halo would span several ranks and the
    replicated path is used).  Extents must be even like the U-Net requires."""
    best = None
    for axis, extent in ((1, d), (2, h)):
        S = slab_size(extent, world)
        if extent % 2 == 0 and S >= halo and (best is None or S > best[1]):
            best = (axis, S)
    return best


def reduce_to_slab(share: torch.Tensor, axis: int, group=None, halo: int = FUSE_HALO):
    """Every rank holds an additive 16-bit ``share`` [n,d,h,w,c] of one volume (sum over ranks = the volume).  Returns
    ``(ext, lo, a, b)``: rank r owns units [a, b) of ``axis`` and gets the SUMMED volume on [lo, lo + ext.shape[axis]) =
    [a - halo, b + halo) clipped to the volume, contiguous in the original layout (b == a on a tail rank without valid units:
    ext is None).

    Two steps, both in the storage format (the 16-bit payload SURVEY 8e budgets):
      1. ``reduce_scatter_tensor`` of the per-rank slabs (payload (world-1)/world of the volume per rank);
      2. the first / last ``halo`` units of the reduced slab go to the two neighbours (point-to-point, 2 * halo / extent of the
         volume) -- recomputing the U-Net on that overlap replaces a per-layer halo exchange."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    E = share.shape[axis]
    S = slab_size(E, world)
    if S < halo:
        raise ValueError(f"reduce_to_slab: slabs of {S} units are thinner than the halo ({halo}); use the replicated path")
    xs = share.movedim(axis, 0)                                    # [E, ...] (a view; contiguous already when n == 1, axis == 1)
    if world * S == E:
        buf = xs.contiguous()
    else:
        buf = torch.zeros((world * S,) + tuple(xs.shape[1:]), dtype=share.dtype, device=share.device)
        buf[:E] = xs
    own = torch.empty((S,) + tuple(xs.shape[1:]), dtype=share.dtype, device=share.device)
    dist.reduce_scatter_tensor(own, buf, group=group)
    valid = [max(0, min(S, E - r * S)) for r in range(world)]
    a, b = rank * S, rank * S + valid[rank]
    if valid[rank] == 0:
        return None, a, a, a
    own = own[:valid[rank]]
    # halo exchange with the neighbours that own valid units (only trailing ranks can be empty)
    n_lo = min(halo, valid[rank - 1]) if rank > 0 else 0
    n_hi = min(halo, valid[rank + 1]) if rank + 1 < world else 0
    lo_buf = torch.empty((n_lo,) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device) if n_lo else None
    hi_buf = torch.empty((n_hi,) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device) if n_hi else None
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    sends, recvs = [], []
    if rank > 0:                                                   # (valid[rank] > 0 implies valid[rank - 1] == S)
        sends.append((own[:min(halo, valid[rank])].contiguous(), peer(rank - 1)))
        recvs.append((lo_buf, peer(rank - 1)))
    if n_hi:
        sends.append((own[valid[rank] - min(halo, valid[rank]):].contiguous(), peer(rank + 1)))
        recvs.append((hi_buf, peer(rank + 1)))
    exchange(sends, recvs, group)
    parts = [t for t in (lo_buf, own, hi_buf) if t is not None]
    ext = (torch.cat(parts, dim=0) if len(parts) > 1 else own).movedim(0, axis).contiguous()
    return ext, a - n_lo, a, b


def exchange(sends, recvs, group=None) -> None:
    """Point-to-point exchange: ``sends`` = [(tensor, global peer rank)], ``recvs`` = [(buffer, global peer rank)], all in one
    ``batch_isend_irecv`` (RCCL groups them into one launch: every pair of GPUs has its own xGMI link, so neighbours exchange
    at link rate).  RCCL takes device pointers and runs on the current stream.  Any other backend (gloo in the one-GPU tests)
    gets HOST copies: ProcessGroupGloo's send / recv hand the raw pointer to a CPU transport that is not ordered with the HIP
    stream -- with device tensors it reads slabs the producing kernel has not finished (seen as 1e-3..1e-2 depth errors at
    configuration 5) -- so the payload goes through ``.cpu()`` (which synchronises) and back."""
    if not sends and not recvs:
        return
    if dist.get_backend(group) == "nccl":
        ops = [dist.P2POp(dist.isend, t, p, group) for t, p in sends] + [dist.P2POp(dist.irecv, t, p, group) for t, p in recvs]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return
    host_s = [(t.cpu(), p) for t, p in sends]
    host_r = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
    ops = [dist.P2POp(dist.isend, t, p, group) for t, p in host_s] + [dist.P2POp(dist.irecv, t, p, group) for t, p in host_r]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for (dst, _), (src, _) in zip(recvs, host_r):
        dst.copy_(src)


def gather_rows(maps: torch.Tensor, rows: int, S: int, group=None) -> torch.Tensor:
    """Row-slab results [n,k,valid,w] of every rank (valid = 0 on a tail rank without rows) -> the full maps [n,k,rows,w] on
    every rank: one all-gather of S-row blocks (tail blocks zero-padded)."""
    world = dist.get_world_size(group)
    n, k, v, w = maps.shape
    blk = torch.zeros((n, k, S, w), dtype=maps.dtype, device=maps.device)
    blk[:, :, :v] = maps
    bufs = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(bufs, blk, group=group)
    return torch.cat(bufs, dim=2)[:, :, :rows].contiguous()


class CollectiveTrace:
    """Measurement aid (bench.py ``sharded`` legs): while active, every ``torch.distributed`` collective the sharded models issue
    (``all_reduce``, ``all_gather``, ``broadcast``) is bracketed by CUDA events on the current stream and recorded as
    ``(name, payload bytes of this rank, milliseconds)``.  ``summary()`` groups the records by (name, bytes).  The events add a
    few microseconds per call; the traced pass is therefore separate from the timed one."""

    NAMES = ("all_reduce", "all_gather", "broadcast", "reduce_scatter_tensor", "all_gather_into_tensor", "batch_isend_irecv")

    def __init__(self):
        self.records = []
        self._saved = {}

    def _wrap(self, name, fn):
        def traced(*args, **kwargs):
            if name == "batch_isend_irecv":         # a list of P2POp: the payload is what this rank SENDS
                tensors = [op.tensor for op in args[0]]
                payload = sum(op.tensor.numel() * op.tensor.element_size() for op in args[0] if op.op is dist.isend)
            else:
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
