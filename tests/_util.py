"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def golden_rig(g) -> str:
    """Camera rig a golden file was generated with (`synthetic.make_cameras(rig=...)`; files older than round 4: "probe")."""
    return str(g["rig"]) if "rig" in g else "probe"


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 tensor with values rounded to bf16 (what the engine stores)."""
    return x.to(torch.bfloat16).to(torch.float32)


def err_stats(got: torch.Tensor, ref: torch.Tensor):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    diff = (got - ref).abs()
    k = int(diff.argmax())
    idx = np.unravel_index(k, tuple(ref.shape)) if ref.numel() else ()
    return {
        "max_abs": float(diff.max()),
        "mean_abs": float(diff.mean()),
        "ref_max": float(ref.abs().max()),
        "ref_mean": float(ref.abs().mean()),
        "rel_l2": float((got - ref).norm() / (ref.norm() + 1e-30)),
        "rel_l1": float(diff.mean() / (ref.abs().mean() + 1e-30)),
        "argmax": tuple(int(i) for i in idx),
        "got_at": float(got.flatten()[k]),
        "ref_at": float(ref.flatten()[k]),
        "nan": int(torch.isnan(got).sum()),
    }


def check_close(name, got, ref, *, max_abs=None, rel_l2=None, rel_l1=None):
    s = err_stats(got, ref)
    line = (f"[parity] {name}: max_abs={s['max_abs']:.3e} (ref max {s['ref_max']:.3e}) rel_l2={s['rel_l2']:.3e} "
            f"rel_l1={s['rel_l1']:.3e} worst@{s['argmax']} got={s['got_at']:.6g} ref={s['ref_at']:.6g} nan={s['nan']}")
    print(line, flush=True)
    assert s["nan"] == 0, line
    if max_abs is not None:
        assert s["max_abs"] <= max_abs, line
    if rel_l2 is not None:
        assert s["rel_l2"] <= rel_l2, line
    if rel_l1 is not None:
        assert s["rel_l1"] <= rel_l1, line
    return s


RETRIES = []      # (test name, reason) of every infrastructure retry of this session: printed in the pytest summary (tests/conftest.py)


def retry_infra(fn):
    """Two ranks sharing one GPU box: a rendezvous / spawn hiccup (a result queue that stays empty, a rank process that dies before it
    reports) is retried ONCE after a pause -- the driver runs this suite with -x on a shared box, where one such hiccup was seen in ~10
    full runs.  Every retry is recorded in ``RETRIES`` and listed in the pytest summary; a second failure of the same kind in a row is
    NOT retried (an intermittent device fault in a shared-GPU test looks exactly like "a rank process exited with code ...": it must
    not be swallowed).  Numerical assertions are never retried."""
    import functools
    import queue as _queue
    import time as _time

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except _queue.Empty as e:
            reason = f"empty result queue ({e!r})"
        except AssertionError as e:
            if "exited with code" not in str(e):
                raise
            reason = str(e)
        RETRIES.append((fn.__name__, reason[:300]))
        print(f"[dist test] infrastructure failure in {fn.__name__}, ONE retry after 10 s: {reason}", flush=True)
        _time.sleep(10)
        return fn(*a, **k)        # a second failure propagates
    return wrapper
