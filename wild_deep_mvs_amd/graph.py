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


# ---- replay INSIDE the mirrors' forward(): the unchanged caller ------------------------------------------------------------------
# The reference's callers do `model(...)` on a fresh sample per iteration (depthmap_eval.py:106, evaluation/run_depthmaps.py:57,
# models/trainer.py:292-304 under no_grad).  They know nothing about GraphedModel, so the mirrors' eval-mode forward() is itself
# wrapped: the SECOND call with the same input signature and option set captures a graph (one-off shapes never pay for a
# capture), later calls replay it.  Opt out per model with `net.graph_replay = False`.
import functools
import operator
import sys
import threading
from collections import OrderedDict

import weakref

_REPLAY = weakref.WeakKeyDictionary()      # model -> replay state
_SIMPLE = (int, float, str, bool, type(None), torch.dtype)
MAX_GRAPHS_PER_MODEL = 3          # captured signatures kept per model (least recently used goes first): a graph pins the
                                  # activation memory of one forward


def _simple(v) -> bool:
    return isinstance(v, _SIMPLE) or (isinstance(v, (list, tuple)) and all(_simple(x) for x in v))


class ReplayHooks:
    """Mixin of the mirrors' top-level modules: counts the events that can replace parameter storage or the module tree
    (``.to()`` / ``.cuda()`` / ``.half()`` go through ``_apply``), so that the per-call keys below can work on cached lists.
    In-place updates (optimizer steps, ``load_state_dict`` -- also through a DataParallel wrapper) are seen through the
    tensors' version counters; ``param.data = ...`` needs ``wild_deep_mvs_amd.invalidate()`` as everywhere in the package."""

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_replay_gen"] = self.__dict__.get("_replay_gen", 0) + 1
        return super()._apply(fn, *args, **kwargs)


_VERSION = operator.attrgetter("_version")


def _model_lists(model: nn.Module, state: dict):
    """(parameters + buffers, non-stock sub-modules) of the model, cached per ``_replay_gen``: walking the module tree costs
    ~0.3 ms for Vis-MVSNet, reading 364 version counters from a cached list ~30 us."""
    gen = model.__dict__.get("_replay_gen", 0)
    if state.get("gen") != gen:
        state["tensors"] = list(model.parameters()) + list(model.buffers())
        state["ptrs"] = tuple(t.data_ptr() for t in state["tensors"])
        state["custom"] = [(name, m) for name, m in model.named_modules() if not type(m).__module__.startswith("torch.nn")]
        state["gen"] = gen
    return state["tensors"], state["custom"]


def _fast_weights_key(model: nn.Module, state: dict):
    from . import ops
    tensors, _ = _model_lists(model, state)
    return (ops.weights_epoch(), state["gen"], state["ptrs"], tuple(map(_VERSION, tensors)))


def _options_key(model: nn.Module, state: dict = None):
    """Every plain option attribute of the model tree (num_depth, depth_nums, interval_scales, nscale, storage_dtype,
    feature_engine, ...) outside the stock torch.nn layers: a graph captured under other options is not replayed.  Returns None
    (= do not replay) when a torch.distributed group is attached (the sharded paths issue collectives: eager only)."""
    custom = _model_lists(model, state)[1] if state is not None else \
        [(name, m) for name, m in model.named_modules() if not type(m).__module__.startswith("torch.nn")]
    key = []
    for name, m in custom:
        for k, v in m.__dict__.items():
            if k.startswith("_") or k == "training":
                continue
            if v is not None and k.endswith("_group"):
                return None
            if isinstance(v, _SIMPLE):
                key.append((name, k, v))
            elif isinstance(v, (list, tuple)) and _simple(v):
                key.append((name, k, repr(v)))
    return tuple(key)


def _has_cpu_tensor(x) -> bool:
    if isinstance(x, torch.Tensor):
        return not x.is_cuda
    if isinstance(x, (list, tuple)):
        return any(_has_cpu_tensor(v) for v in x)
    if isinstance(x, dict):
        return any(_has_cpu_tensor(v) for v in x.values())
    return False


def _requires_grad(x) -> bool:
    if isinstance(x, torch.Tensor):
        return x.requires_grad
    if isinstance(x, (list, tuple)):
        return any(_requires_grad(v) for v in x)
    if isinstance(x, dict):
        return any(_requires_grad(v) for v in x.values())
    return False


MAX_EVICTIONS_PER_SIGNATURE = 2     # a signature the LRU dropped this often stays eager (alternating > MAX_GRAPHS_PER_MODEL shapes)


def replayable(forward):
    """Decorator of a mirror's ``forward``: in eval mode, for CUDA inputs, WITH AUTOGRAD OFF (`torch.no_grad()` /
    `inference_mode`, as the reference's eval loops run it -- depthmap_eval.py:100-106, evaluation/run_depthmaps.py:53-57), replay a
    hipGraph of the forward from the second call of a signature on.  Falls through to the eager forward in train() mode, whenever
    autograd is enabled (an eval-mode call under grad returns a graph-connected result on EVERY call, not only the first), under an
    outer capture (GraphedModel, bench.py), with ``taps=`` (a dict the caller wants filled), with CPU inputs, with a
    torch.distributed group attached, and when
    ``self.graph_replay`` is False.  The key also holds the autocast state.  A signature evicted MAX_EVICTIONS_PER_SIGNATURE times
    by the LRU (more alternating input shapes than MAX_GRAPHS_PER_MODEL, e.g. a mixed-resolution eval set) is not captured
    again: a capture costs three forwards.  Eager fall-backs run OUTSIDE the model's replay lock."""

    @functools.wraps(forward)
    def wrapper(self, *args, **kwargs):
        if (self.training or not getattr(self, "graph_replay", True) or kwargs.get("taps") is not None
                or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing()
                or torch.is_grad_enabled()
                or _has_cpu_tensor(args) or _has_cpu_tensor(kwargs)):
            return forward(self, *args, **kwargs)
        state = _REPLAY.get(self)        # (kept outside the module: a lock and HIP graphs must not be deep-copied / pickled with it)
        if state is None:
            state = _REPLAY[self] = {"lock": threading.Lock(), "seen": OrderedDict(), "graphs": OrderedDict(), "failed": set(),
                                     "evicted": {}}
        okey = _options_key(self, state)
        if okey is None:
            return forward(self, *args, **kwargs)
        # + the tuning-knob generation and the upper-case switches of the model's own module (models.MVSNet.model.FUSED_TAIL)
        from . import _lib
        flags = tuple((k, v) for k, v in sys.modules[type(self).__module__].__dict__.items() if k.isupper() and isinstance(v, _SIMPLE))
        autocast = (torch.is_autocast_enabled(), str(torch.get_autocast_gpu_dtype()) if torch.is_autocast_enabled() else "")
        key = (_sig(args), _sig(kwargs), okey, torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, _lib.TUNING_GEN,
               flags, autocast)
        eager = False
        with state["lock"]:
            wkey = None
            entry = None
            if key in state["failed"]:
                eager = True
            else:
                wkey = _fast_weights_key(self, state)
                entry = state["graphs"].get(key)
                if entry is not None and entry[3] != wkey:                     # weights changed since the capture
                    del state["graphs"][key]
                    entry = None
                if entry is None and key not in state["seen"]:                  # first sight of this signature: eager
                    state["seen"][key] = True
                    while len(state["seen"]) > 64:
                        state["seen"].popitem(last=False)
                    eager = True
            if not eager and entry is None:
                static_args, static_kwargs = _clone_static(args), _clone_static(kwargs)
                try:
                    with torch.no_grad():
                        side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            forward(self, *static_args, **static_kwargs)     # caches, workspaces, library handles
                        torch.cuda.current_stream().wait_stream(side)
                        torch.cuda.synchronize()
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                            static_out = forward(self, *static_args, **static_kwargs)
                except Exception:                                           # not capturable here: stay eager for this signature
                    state["failed"].add(key)
                    torch.cuda.synchronize()
                    eager = True
                else:
                    entry = (graph, (static_args, static_kwargs), static_out, wkey)
                    state["graphs"][key] = entry
                    while len(state["graphs"]) > MAX_GRAPHS_PER_MODEL:
                        old_key, _ = state["graphs"].popitem(last=False)
                        n = state["evicted"][old_key] = state["evicted"].get(old_key, 0) + 1
                        if n >= MAX_EVICTIONS_PER_SIGNATURE:
                            state["failed"].add(old_key)                    # thrashing: this signature stays eager from now on
            elif not eager:
                state["graphs"].move_to_end(key)
            if not eager:
                graph, (static_args, static_kwargs), static_out, _ = entry
                _copy_in(static_args, args)
                _copy_in(static_kwargs, kwargs)
                graph.replay()
                return _clone_out(static_out)
        return forward(self, *args, **kwargs)                                # eager fall-backs run without the lock

    wrapper.eager = forward
    return wrapper


def drop_replay_graphs(model: nn.Module) -> None:
    """Forget the graphs captured inside ``model``'s forward (frees their memory pools)."""
    _REPLAY.pop(model, None)


# ---- a batch of independent reference views as free-running per-view graphs -----------------------------------------------------
class ViewPipeline:
    """The MVSNet hot path over a batch of B independent reference views, one single-branch hipGraph PER VIEW, each replayed on its own
    HIP stream; ``step()`` launches one replay of every view and returns at once, and consecutive steps are NOT joined: a view's
    stream runs ahead of the others, so the views drift out of phase and one view's warp runs beside another view's U-Net instead of
    three warps, then three conv0s ... in lockstep.  ``results()`` joins the view streams into the caller's stream and returns
    (depth [B,h,w], confidence [B,h,w]) of the LAST step.

    Why not one graph with B parallel branches (``MVSNet._hot_path_streams`` under capture, rounds 3-5's step): measured in round 6
    (profiles/r06_step_schedule.txt; rocprofv3 timelines) the third branch of such a graph starts ~0.46 ms after the first two on
    ROCm 7.2, its U-Net then runs alone at the end of the step, and every replay ends with a join: 0.918-0.927 ms per 3-view step
    against 0.894-0.907 ms for the free-running per-view graphs on the same boxes, bit-identical outputs.  The reference has no
    counterpart (it runs one sample at a time, depthmap_eval.py:100-106).

    Inputs are read where they lie (``features_cl`` [V x [B,h,w,C]], ``proj`` [B,V,4,4], ``depth_values`` [B,D]): write new inputs
    into these tensors on the caller's stream before ``step()`` (every view stream waits for the caller's stream first)."""

    def __init__(self, net, features_cl, proj, depth_values, reference_frame: int = 0, warmup: int = 2):
        if net.training:
            raise RuntimeError("ViewPipeline replays the eval-mode hot path")
        self.net = net
        self.B = int(features_cl[0].shape[0])
        dev = features_cl[0].device
        self.device = dev
        depth_values = depth_values.to(torch.float32)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.B)]
        self.graphs, self.outs = [], []
        self._inputs = (features_cl, proj, depth_values)           # keep the static inputs alive
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for b in range(self.B):
                fb = [f[b:b + 1] for f in features_cl]              # contiguous views of one batch item: replays read the live tensors
                pb, db = proj[b:b + 1], depth_values[b:b + 1]
                st = self.streams[b]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    for _ in range(max(1, warmup)):                 # packed-weight caches, dynamic-LDS attributes
                        net.hot_path(fb, pb, db, reference_frame)
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                    out = net.hot_path(fb, pb, db, reference_frame)
                self.graphs.append(g)
                self.outs.append(out)
        torch.cuda.synchronize(dev)

    def step(self) -> None:
        """One replay of every view's graph on its own stream; returns without waiting for the GPU."""
        cur = torch.cuda.current_stream(self.device)
        for st, g in zip(self.streams, self.graphs):
            st.wait_stream(cur)                                     # inputs written on the caller's stream are visible to the view
            with torch.cuda.stream(st):
                g.replay()

    def results(self):
        """Join: the caller's stream waits for every view stream; (depth, confidence) of the last step, as new tensors."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)
        return torch.cat([o[0] for o in self.outs], 0), torch.cat([o[1] for o in self.outs], 0)
