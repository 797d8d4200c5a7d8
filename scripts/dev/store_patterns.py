#!/usr/bin/env python3
"""Dev (round 4): time the four store patterns of scripts/ubench/store_patterns.hip on the cost volume's 251 MB."""
import ctypes as C
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "ubench", "libsp.so"))
lib.sp_run.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
out = torch.empty(3932160 * 64, dtype=torch.uint8, device="cuda")
tiles = 3932160 // 64
st = torch.cuda.current_stream().cuda_stream


def time_us(pat, blocks, spread, steps=30, warm=5):
    for _ in range(warm):
        lib.sp_run(pat, out.data_ptr(), tiles, blocks, spread, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        lib.sp_run(pat, out.data_ptr(), tiles, blocks, spread, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


names = {0: "lane = voxel, 64-byte stride", 1: "contiguous", 2: "pieces of a voxel in lanes c, c+16, c+32, c+48", 3: "quad of lanes = voxel"}
for blocks in (768, 3072):
    for spread in (0, 200):
        for pat in (0, 1, 2, 3):
            ts = sorted(time_us(pat, blocks, spread) for _ in range(3))
            print(f"blocks {blocks} spread {spread:3d} pattern {pat} ({names[pat]}): {ts[1]:.1f} us = {251.66 / ts[1] * 1e3:.0f} GB/s", flush=True)
