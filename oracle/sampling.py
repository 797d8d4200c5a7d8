"""First-principles bilinear gather (numpy).  TEST INFRASTRUCTURE ONLY.

Pins the one ATen primitive the whole plane sweep rests on:
``F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True)`` as
called at reference ``models/MVSNet/module.py:165``,
``models/CVP_MVSNet/models/modules.py:124,278`` and
``models/VisMVSNet/homography.py:101``.  With ``align_corners=True`` a
normalised coordinate ``g`` addresses pixel index ``(g + 1) / 2 * (size - 1)``;
the four neighbours are blended with weights given by the distances to the
opposite corner and any neighbour outside the image contributes zero.
"""
from __future__ import annotations

import numpy as np


def unnormalize(g: np.ndarray, size: int) -> np.ndarray:
    """normalised [-1, 1] -> pixel index, align_corners=True."""
    return (g.astype(np.float32) + np.float32(1.0)) / np.float32(2.0) * np.float32(size - 1)


def bilinear_zero_pad(img: np.ndarray, ix: np.ndarray, iy: np.ndarray) -> np.ndarray:
    """img [C, H, W]; ix, iy arrays of pixel indices (any shape) -> [C, *ix.shape].

    Plain loops over the four taps; meant for small cases only.
    """
    C, H, W = img.shape
    ix = ix.astype(np.float32)
    iy = iy.astype(np.float32)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    out = np.zeros((C,) + ix.shape, dtype=np.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            xn = x0 + dx
            yn = y0 + dy
            # weight = area of the rectangle spanned with the opposite corner
            wx = (x0 + 1 - ix) if dx == 0 else (ix - x0)
            wy = (y0 + 1 - iy) if dy == 0 else (iy - y0)
            inside = (xn >= 0) & (xn <= W - 1) & (yn >= 0) & (yn <= H - 1)
            xi = np.clip(xn, 0, W - 1).astype(np.int64)
            yi = np.clip(yn, 0, H - 1).astype(np.int64)
            tap = img[:, yi, xi]
            out += tap * (wx * wy * inside).astype(np.float32)[None]
    return out
