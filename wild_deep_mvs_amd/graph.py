"""hipGraph replay of a model's eval-mode ``forward()``.

The engine's forward is a few hundred short launches (Vis-MVSNet at 512x640: ~300, most of them 5-30 us), so in eager
mode the host's launch rate, not the GPU, bounds the latency.  ``GraphedModel`` captures one forward per input signature
(shapes, dtypes, keyword arguments) into a HIP graph -- every pscv entry point launches asynchronously on the current
stream and allocates nothing itself, and the PyTorch plumbing in between (layout conversions, the 2-D UncertNet of
Vis-MVSNet) is capturable -- and replays it on later calls: inputs are copied into the graph's static buffers, outputs are
returned as fresh copies.  The reference has no counterpart (it runs eager PyTorch); its callers use the wrapper like the
module itself:

    net = GraphedModel(Frontend().cuda().eval())
    out = net(imgs, K, R, t, depth_min, depth_max)         # first call per signature: warm-up + capture, then replay

Inference only.  Tensors in keyword arguments are copied into static buffers like the positional ones.  Weights are read at
capture time: a graph is re-captured when a parameter / buffer changed (address or version; ``.data`` writes need
``wild_deep_mvs_amd.invalidate()``); call ``reset()`` after changing ``storage_dtype`` or another engine option.
"""
from __future__ import annotations

from typing import Any, Dict, Tuple

import torch
import torch.nn as nn


def _sig(x) -> Any:
    if isinstance(x, torch.Tensor):
        return ("T", tuple(x.shape), str(x.dtype), str(x.device))
    if isinstance(x, (list, tuple)):
        return (type(x).__name__,) + tuple(_sig(v) for v in x)
    if isinstance(x, dict):
        return ("D",) + tuple((k, _sig(v)) for k, v in sorted(x.items()))
    return ("V", repr(x))


def _clone_static(x):
    if isinstance(x, torch.Tensor):
        return x.detach().clone()
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_static(v) for v in x)
    if isinstance(x, dict):
        return {k: _clone_static(v) for k, v in x.items()}
    return x


def _copy_in(dst, src):
    if isinstance(dst, torch.Tensor):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_in(d, s)
    elif isinstance(dst, dict):
        for k, d in dst.items():
            _copy_in(d, src[k])


def _weights_key(model: nn.Module):
    """(address, version) of every parameter and buffer + the package's cache generation: a graph captured against other
    weights is not replayed (in-place updates and load_state_dict bump the version; ``.data`` writes need
    ``wild_deep_mvs_amd.invalidate()``, which bumps the generation)."""
    from . import ops
    return (ops.weights_epoch(),) + tuple((t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))


def _clone_out(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_out(v) for v in x)
    if isinstance(x, dict):
        return {k: _clone_out(v) for k, v in x.items()}
    return x


class GraphedModel(nn.Module):
    def __init__(self, model: nn.Module, warmup: int = 2):
        super().__init__()
        self.model = model
        self.warmup = int(warmup)
        self._graphs: Dict[Any, tuple] = {}

    def reset(self):
        """Drop every captured graph (after the weights or a storage / engine option of the model changed)."""
        self._graphs.clear()

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("model"), name)

    def forward(self, *args, **kwargs):
        if self.model.training:
            raise RuntimeError("GraphedModel replays an eval-mode forward; call .eval() (training runs eagerly on the module itself)")
        key = (_sig(args), _sig(kwargs))
        wkey = _weights_key(self.model)
        entry = self._graphs.get(key)
        if entry is not None and entry[3] != wkey:     # the weights changed since the capture
            entry = None
        if entry is None:
            # tensors anywhere in the positional AND keyword arguments (lists, tuples, dicts) get static buffers
            static_args, static_kwargs = _clone_static(args), _clone_static(kwargs)
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(self.warmup):      # builds the packed-weight caches, workspaces and library handles
                        self.model(*static_args, **static_kwargs)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static_out = self.model(*static_args, **static_kwargs)
            entry = (graph, (static_args, static_kwargs), static_out, wkey)
            self._graphs[key] = entry
        graph, (static_args, static_kwargs), static_out, _ = entry
        _copy_in(static_args, args)
        _copy_in(static_kwargs, kwargs)
        graph.replay()
        return _clone_out(static_out)
