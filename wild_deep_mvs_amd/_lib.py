"""ctypes binding of libpscv.so (the C ABI declared in include/pscv.h).

The library is built in-tree by ``wild_deep_mvs_amd/csrc/Makefile`` (``__graft_entry__.build()``)
and must be present: there is no PyTorch / CPU fallback for the hot path.  ``lib()`` raises
``PscvMissingError`` when the shared object cannot be loaded.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSCV_LIB") or os.path.join(_HERE, "libpscv.so")     # (PSCV_LIB: A/B builds of scripts/dev)
CSRC = os.path.join(_HERE, "csrc")

# mirror of include/pscv.h
ABI_VERSION = 9
F32, BF16, F16 = 0, 1, 2
GEOM_PROJ, GEOM_HOMOG = 0, 1
COST_VARIANCE, COST_VARIANCE_CVP, COST_SOFTMIN, COST_GROUPCORR, COST_WARP_ONLY, COST_VARIANCE_PARTIAL = 0, 1, 2, 3, 4, 5
CONV_S1, CONV_S2, CONV_T2, CONV_S1P8, CONV_S1C1, CONV_T2P8 = 0, 1, 2, 3, 4, 5
EPI_RELU_PRE, EPI_RELU_POST = 1, 2
MAX_SRC = 16
CAM_FLOATS = 18
GEO_MAX_SRC = 32
GEO_CAM_FLOATS = 30

EXPORTS = ("pscv_last_error", "pscv_abi_version", "pscv_set_tuning", "pscv_proj_cams", "pscv_homog_cams", "pscv_warp_cost",
           "pscv_fuse_pairs", "pscv_fuse_finish", "pscv_geo_filter", "pscv_pack_conv2d_weights", "pscv_conv2d",
           "pscv_pack_conv3d_weights", "pscv_conv3d", "pscv_softargmin", "pscv_train_workspace_floats", "pscv_bn_stats",
           "pscv_bn_act", "pscv_bn_bwd_reduce", "pscv_bn_bwd_apply", "pscv_softargmin_bwd", "pscv_conv3d_wgrad_workspace",
           "pscv_conv3d_wgrad", "pscv_warp_cost_bwd", "pscv_cvp_depth_hypos", "pscv_relu_bwd", "pscv_fuse_pairs_bwd", "pscv_pack_conv3d_weights_device", "pscv_conv2d_ex", "pscv_variance_finish", "pscv_softargmin_window",
           "pscv_photo_warp", "pscv_photo_warp_bwd", "pscv_ssim", "pscv_ssim_bwd", "pscv_bn_finalize", "pscv_bn_bwd_coeffs", "pscv_cvp_cams", "pscv_homography_warp", "pscv_homography_warp_bwd", "pscv_prob_softargmin", "pscv_prob_softargmin_workspace",
           "pscv_set_tuning_thread", "pscv_get_tuning", "pscv_conv3d_cat2", "pscv_uncert_net", "pscv_head_index_entropy", "pscv_image_prep", "pscv_conv3d_block8",
           "pscv_bn_stats_grouped", "pscv_bn_finalize_grouped", "pscv_bn_act_grouped", "pscv_bn_bwd_reduce_grouped", "pscv_bn_bwd_coeffs_grouped",
           "pscv_bn_bwd_apply_grouped", "pscv_pack_conv2d_weights_device", "pscv_leaky_relu_bwd", "pscv_leaky_relu_bwd_sum", "pscv_pack_conv2d_weights_device_ex", "pscv_warp_cost_rows",
           "pscv_tail_sweep", "pscv_tail_sweep_workspace")


class PscvMissingError(RuntimeError):
    pass


class PscvError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


STAMP_PATH = os.path.join(CSRC, ".build_stamp")


def source_hash() -> str:
    """sha256 over every file the library is compiled from (csrc/*.hip|h|cpp, the Makefile, include/pscv.h)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile")
    for path in [os.path.join(CSRC, f) for f in files] + [os.path.join(os.path.dirname(_HERE), "include", "pscv.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile libpscv.so for gfx950 with hipcc (cross-compiles without a GPU).  File times do not survive a copy of the tree, so
    the decision is by CONTENT: the sources' hash is stored next to the objects (csrc/.build_stamp); a library built from other
    sources (or none) is rebuilt from scratch with `make -B`, otherwise `make` only links what is missing.  Prints which happened."""
    want = source_hash()
    have = open(STAMP_PATH).read().strip() if os.path.exists(STAMP_PATH) else ""
    fresh = force or have != want or not os.path.exists(LIB_PATH)
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if fresh else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libpscv.so failed")
    with open(STAMP_PATH, "w") as fh:
        fh.write(want + "\n")
    print(f"[pscv build] {'full rebuild (make -B): sources changed or no stamp' if fresh else 'up to date: sources match the stamp of the built library'}"
          f" [{want[:12]}]", flush=True)
    return LIB_PATH


def _declare(lib):
    vp, i, l, f = C.c_void_p, C.c_int, C.c_long, C.c_float
    lib.pscv_last_error.restype = C.c_char_p
    lib.pscv_last_error.argtypes = []
    lib.pscv_abi_version.restype = i
    lib.pscv_abi_version.argtypes = []
    lib.pscv_set_tuning.restype = i
    lib.pscv_set_tuning.argtypes = [C.c_char_p, i]
    lib.pscv_set_tuning_thread.restype = i
    lib.pscv_set_tuning_thread.argtypes = [C.c_char_p, i, i]
    lib.pscv_get_tuning.restype = i
    lib.pscv_get_tuning.argtypes = [C.c_char_p, C.POINTER(i)]
    lib.pscv_proj_cams.restype = i
    lib.pscv_proj_cams.argtypes = [vp, i, i, i, vp, vp]
    lib.pscv_homog_cams.restype = i
    lib.pscv_homog_cams.argtypes = [vp, vp, i, i, f, vp, vp]
    lib.pscv_fuse_pairs.restype = i
    lib.pscv_fuse_pairs.argtypes = [C.POINTER(vp), C.POINTER(vp), i, i, vp, vp, i, i, i, i, i, vp]
    lib.pscv_fuse_finish.restype = i
    lib.pscv_fuse_finish.argtypes = [vp, vp, i, vp, i, i, i, i, vp]
    lib.pscv_pack_conv2d_weights.restype = l
    lib.pscv_pack_conv2d_weights.argtypes = [vp, i, i, i, i, i, vp]
    lib.pscv_conv2d.restype = i
    lib.pscv_conv2d.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, vp]
    lib.pscv_bn_finalize.restype = i
    lib.pscv_bn_finalize.argtypes = [vp, l, i, vp, vp, f, f, vp, vp, vp, vp, vp]
    lib.pscv_bn_bwd_coeffs.restype = i
    lib.pscv_bn_bwd_coeffs.argtypes = [vp, vp, vp, vp, l, i, vp, vp]
    lib.pscv_photo_warp.restype = i
    lib.pscv_photo_warp.argtypes = [vp] * 10 + [i, i, i, i, i, vp]
    lib.pscv_photo_warp_bwd.restype = i
    lib.pscv_photo_warp_bwd.argtypes = [vp] * 6 + [i, i, i, i, i, vp]
    lib.pscv_ssim.restype = i
    lib.pscv_ssim.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
    lib.pscv_ssim_bwd.restype = i
    lib.pscv_ssim_bwd.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
    lib.pscv_softargmin_window.restype = i
    lib.pscv_softargmin_window.argtypes = [vp, vp, vp, f, i, i, i, i, i, vp]
    lib.pscv_variance_finish.restype = i
    lib.pscv_variance_finish.argtypes = [vp, l, i, i, i, vp, vp]
    lib.pscv_conv2d_ex.restype = i
    lib.pscv_conv2d_ex.argtypes = [vp, i, vp, vp, vp, vp, i, i, vp, i, i, i, i, i, i, i, i, i, i, i, f, vp]
    lib.pscv_geo_filter.restype = i
    lib.pscv_geo_filter.argtypes = [vp, C.POINTER(vp), C.POINTER(i), i, vp, i, i, f, f, f, i, vp, vp, vp, vp, vp]
    lib.pscv_warp_cost.restype = i
    lib.pscv_warp_cost.argtypes = [vp, C.POINTER(vp), i, vp, vp, l, i, i, i, f, vp, i, i, i, i, i, i, i, i, i, vp]
    lib.pscv_warp_cost_rows.restype = i
    lib.pscv_warp_cost_rows.argtypes = [vp, C.POINTER(vp), i, vp, vp, l, i, i, i, f, vp, i, i, i, i, i, i, i, i, i, i, vp]
    lib.pscv_pack_conv3d_weights.restype = l
    lib.pscv_pack_conv3d_weights.argtypes = [vp, i, i, i, i, i, vp]
    lib.pscv_pack_conv3d_weights_device.restype = i
    lib.pscv_pack_conv3d_weights_device.argtypes = [vp, i, i, i, i, i, vp, vp]
    lib.pscv_conv3d.restype = i
    lib.pscv_conv3d.argtypes = [vp, i, i, i, vp, vp, vp, vp, vp, i, i, vp, i, i, i, i, i, i, i, i, i, i, i, vp]
    lib.pscv_conv3d_cat2.restype = i
    lib.pscv_conv3d_cat2.argtypes = [vp, i, i, vp, i, i, i, vp, vp, vp, vp, vp, i, i, vp, i, i, i, i, i, i, i, i, i, vp]
    lib.pscv_uncert_net.restype = i
    lib.pscv_uncert_net.argtypes = [vp, vp, vp, i, i, i, vp]
    lib.pscv_head_index_entropy.restype = i
    lib.pscv_head_index_entropy.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, i, vp, vp, C.c_long, vp, vp, i, i, i, i, vp]
    lib.pscv_image_prep.restype = i
    lib.pscv_image_prep.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp]
    lib.pscv_conv3d_block8.restype = i
    lib.pscv_conv3d_block8.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, vp, vp, vp, vp, i, i, vp, i, i, i, i, i, i, vp]
    lib.pscv_softargmin.restype = i
    lib.pscv_softargmin.argtypes = [vp, i, vp, l, i, vp, vp, vp, vp, vp, vp, i, f, i, i, i, i, i, vp]
    lib.pscv_train_workspace_floats.restype = l
    lib.pscv_train_workspace_floats.argtypes = []
    lib.pscv_leaky_relu_bwd_sum.restype = i
    lib.pscv_leaky_relu_bwd_sum.argtypes = [vp, vp, i, l, i, f, vp, vp, vp, vp]
    lib.pscv_leaky_relu_bwd.restype = i
    lib.pscv_leaky_relu_bwd.argtypes = [vp, vp, i, l, i, f, vp, vp]
    lib.pscv_pack_conv2d_weights_device_ex.restype = i
    lib.pscv_pack_conv2d_weights_device_ex.argtypes = [vp, i, i, i, i, i, i, vp, vp]
    lib.pscv_pack_conv2d_weights_device.restype = i
    lib.pscv_pack_conv2d_weights_device.argtypes = [vp, i, i, i, i, i, vp, vp]
    lib.pscv_bn_stats_grouped.restype = i
    lib.pscv_bn_stats_grouped.argtypes = [vp, i, l, i, i, vp, vp, vp]
    lib.pscv_bn_finalize_grouped.restype = i
    lib.pscv_bn_finalize_grouped.argtypes = [vp, l, i, i, vp, vp, f, f, vp, vp, vp, vp, vp]
    lib.pscv_bn_act_grouped.restype = i
    lib.pscv_bn_act_grouped.argtypes = [vp, i, l, i, i, vp, vp, i, i, vp, vp, vp]
    lib.pscv_bn_bwd_reduce_grouped.restype = i
    lib.pscv_bn_bwd_reduce_grouped.argtypes = [vp, vp, i, l, i, i, vp, vp, i, i, vp, vp, vp]
    lib.pscv_bn_bwd_coeffs_grouped.restype = i
    lib.pscv_bn_bwd_coeffs_grouped.argtypes = [vp, vp, vp, i, vp, l, i, i, vp, vp]
    lib.pscv_bn_bwd_apply_grouped.restype = i
    lib.pscv_bn_bwd_apply_grouped.argtypes = [vp, vp, i, l, i, i, vp, vp, i, i, vp, vp, vp, i, vp, vp]
    lib.pscv_bn_stats.restype = i
    lib.pscv_bn_stats.argtypes = [vp, i, l, i, vp, vp, vp]
    lib.pscv_bn_act.restype = i
    lib.pscv_bn_act.argtypes = [vp, i, l, i, vp, vp, i, vp, vp, vp]
    lib.pscv_bn_bwd_reduce.restype = i
    lib.pscv_bn_bwd_reduce.argtypes = [vp, vp, i, l, i, vp, vp, i, vp, vp, vp]
    lib.pscv_bn_bwd_apply.restype = i
    lib.pscv_bn_bwd_apply.argtypes = [vp, vp, i, l, i, vp, vp, i, vp, vp, vp, vp, vp]
    lib.pscv_softargmin_bwd.restype = i
    lib.pscv_softargmin_bwd.argtypes = [vp, vp, l, i, vp, vp, vp, vp, i, i, i, i, i, vp]
    lib.pscv_relu_bwd.restype = i
    lib.pscv_relu_bwd.argtypes = [vp, vp, i, l, i, vp, vp]
    lib.pscv_fuse_pairs_bwd.restype = i
    lib.pscv_fuse_pairs_bwd.argtypes = [C.POINTER(vp), C.POINTER(vp), i, i, vp, C.POINTER(vp), C.POINTER(vp), i, i, i, i, vp]
    lib.pscv_conv3d_wgrad_workspace.restype = l
    lib.pscv_conv3d_wgrad_workspace.argtypes = [i, i, i, i, i, i, i]
    lib.pscv_conv3d_wgrad.restype = i
    lib.pscv_conv3d_wgrad.argtypes = [vp, i, i, i, vp, i, i, i, i, i, i, i, i, i, vp, vp, i, vp]
    lib.pscv_prob_softargmin_workspace.restype = C.c_long
    lib.pscv_prob_softargmin_workspace.argtypes = [i, i, i, i]
    lib.pscv_prob_softargmin.restype = i
    lib.pscv_prob_softargmin.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, i, vp, C.c_long, vp, vp, C.c_long, vp, vp, i, i, i, i, vp]
    lib.pscv_tail_sweep.restype = i
    lib.pscv_tail_sweep.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, vp, i, i, vp, vp, vp, vp, i, vp, vp, l, vp, l, vp, vp, i, i, i, i, vp]
    lib.pscv_tail_sweep_workspace.restype = C.c_long
    lib.pscv_tail_sweep_workspace.argtypes = [i, i, i, i]
    lib.pscv_homography_warp.restype = i
    lib.pscv_homography_warp.argtypes = [vp, vp, i, vp, i, i, i, i, i, i, vp]
    lib.pscv_homography_warp_bwd.restype = i
    lib.pscv_homography_warp_bwd.argtypes = [vp, vp, i, vp, i, i, i, i, i, i, vp]
    lib.pscv_cvp_cams.restype = i
    lib.pscv_cvp_cams.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp, vp]
    lib.pscv_cvp_depth_hypos.restype = i
    lib.pscv_cvp_depth_hypos.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, vp]
    lib.pscv_warp_cost_bwd.restype = i
    lib.pscv_warp_cost_bwd.argtypes = [vp, C.POINTER(vp), i, vp, vp, l, i, i, i, f, vp, vp, C.POINTER(vp), vp, i, i, i, i, i,
                                       i, i, i, i, vp]


def lib():
    """The loaded library (loads on first use; import torch first so that its HIP runtime is the one
    both sides share -- both carry SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise PscvMissingError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or make -C wild_deep_mvs_amd/csrc). The plane-sweep engine has no CPU / PyTorch fallback.")
            try:
                import torch  # noqa: F401  (loads torch's libamdhip64 first)
            except Exception:  # pragma: no cover
                pass
            try:
                handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
            except OSError as e:
                raise PscvMissingError(f"cannot load {LIB_PATH}: {e}") from e
            _declare(handle)
            ver = handle.pscv_abi_version()
            if ver != ABI_VERSION:
                raise PscvMissingError(f"libpscv.so ABI {ver} != binding ABI {ABI_VERSION}: rebuild")
            _lib = handle
    return _lib


TUNING_GEN = 0      # bumped by every knob change: captured hipGraphs (graph.replayable) froze the kernel selection of their capture


def set_tuning(key: str, value: int) -> None:
    """Process-wide: also seen by launches from other host threads (autograd's backward thread, DataParallel replicas)."""
    global TUNING_GEN
    TUNING_GEN += 1
    check(lib().pscv_set_tuning(key.encode(), int(value)), "pscv_set_tuning")


def get_tuning(key: str) -> int:
    """The value the calling thread's next launch would use."""
    v = C.c_int(0)
    check(lib().pscv_get_tuning(key.encode(), C.byref(v)), "pscv_get_tuning")
    return int(v.value)


def set_tuning_thread(key: str, value: int, enable: bool = True) -> None:
    """Override (or, enable=False, stop overriding) a knob for the calling host thread only."""
    global TUNING_GEN
    TUNING_GEN += 1
    check(lib().pscv_set_tuning_thread(key.encode(), int(value), 1 if enable else 0), "pscv_set_tuning_thread")


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().pscv_last_error().decode("utf-8", "replace")
        raise PscvError(f"{what} failed (rc={rc}): {msg}")
