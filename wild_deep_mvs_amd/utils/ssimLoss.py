"""Drop-in for the reference's ``utils/ssimLoss.py``: ``SSIM()(img1, img2)`` = 1 - SSIM per channel with the 11x11 Gaussian
window (sigma 1.5, zero padding), on the pscv HIP kernels (``pscv_ssim`` / ``pscv_ssim_bwd``; forward and backward)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import training as T


class SSIM(nn.Module):
    """reference utils/ssimLoss.py:47-60.  img1 [n1,3,h,w] is the reference image (data, no gradient); img2 [n1*rep,3,h,w]
    the warped images -- item n of img2 is compared with item n // rep of img1, so one call covers all source views."""

    def __init__(self, window_size: int = 11):
        super().__init__()
        if window_size != 11:
            raise NotImplementedError("pscv SSIM implements the reference's only configuration: window_size=11, sigma=1.5")
        self.window_size, self.channel = window_size, 3

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        return T.SSIMFn.apply(img1.detach().to(torch.float32), img2.to(torch.float32))
