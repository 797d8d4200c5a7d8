#!/usr/bin/env python3
"""Dev (round 4): the self-checking op_sel victims of scripts/ubench/lds_pk_overlap.hip (v_pk_mov_b32 / v_pk_mul_f32 / v_pk_add_f32 with a
high-half selector) beside the engine's real conv0 on a second stream, and alone."""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from wild_deep_mvs_amd import _lib as L, ops  # noqa: E402
from test_gpu_overlap import Soak  # noqa: E402

soak = Soak(L, ops)
lpo = C.CDLL(os.path.join(REPO, "scripts", "ubench", "liblpo_opselhi.so"))
lpo.lpo_opsel_victim.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for partner in (False, True):
    errs = torch.zeros(36, dtype=torch.int32, device="cuda")
    for it in range(60):
        if partner:
            with torch.cuda.stream(soak.sb):
                for _ in range(3):
                    ops.conv3d(soak.px, soak.player)
        with torch.cuda.stream(soak.sa):
            assert lpo.lpo_opsel_victim(errs.data_ptr(), 400, soak.sa.cuda_stream) == 0
        torch.cuda.synchronize()
    e = errs.cpu().tolist()
    names = ["v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_add_f32 op_sel:[0,1]",
             "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]"]
    print(f"conv0 on a second stream: {partner}; wrong results per 16-lane group (lanes 0-15, 16-31, 32-47, 48-63), 60 launches x 1 M threads x 400 rounds:", flush=True)
    for k, nm in enumerate(names):
        print(f"   {nm:32s} {e[4 * k:4 * k + 4]}", flush=True)
