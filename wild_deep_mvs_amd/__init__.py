"""wild_deep_mvs_amd -- MI355X-native plane-sweep cost-volume engine behind the reference's models/* API.

    csrc/ + libpscv.so   hand-written HIP kernels for gfx950 behind a C ABI (include/pscv.h)
    _lib.py / ops.py     ctypes binding and tensor-level wrappers (PyTorch = device memory + streams)
    models/              mirror of the reference's ``models`` package (same import paths / names)
    synthetic.py         seeded cameras, images, features and sharpened weights (no datasets here)
    dist.py              multi-GPU sharding of the path (one process per GPU, RCCL via torch.distributed)
"""
import importlib
import sys

__all__ = ["install_as_models", "invalidate"]


def invalidate() -> None:
    """Drop every packed-weight / folded-BatchNorm cache (see ``ops.invalidate_weight_caches``): required after a write through
    ``parameter.data`` (EMA copies, clamps, old-style optimisers), which does not bump the tensor version the caches key on."""
    from . import ops
    ops.invalidate_weight_caches()


def install_as_models() -> None:
    """Make ``import models.MVSNet.model`` (the reference's import paths, train.py:33-36,
    evaluation/pipeline_utils.py:131-154) resolve to this package's mirror."""
    pkg = importlib.import_module(__name__ + ".models")
    sys.modules["models"] = pkg
    for sub in ("MVSNet", "MVSNet.model", "MVSNet.module", "VisMVSNet", "VisMVSNet.frontend", "VisMVSNet.model_cas",
                "VisMVSNet.nn_utils", "VisMVSNet.homography", "VisMVSNet.preproc", "CVP_MVSNet", "CVP_MVSNet.frontend",
                "CVP_MVSNet.models", "CVP_MVSNet.models.net", "CVP_MVSNet.models.modules", "utils", "trainer"):
        try:
            sys.modules["models." + sub] = importlib.import_module(f"{__name__}.models.{sub}")
        except ImportError:
            pass
    # (``utils.trainer`` / ``utils.utils_3D`` / ``utils.ssimLoss`` are the caller's own harness modules; their counterparts live
    #  in ``wild_deep_mvs_amd.utils`` and are what the mirrors above import, so nothing is registered under ``utils``.)
