"""``scale_camera`` -- the only part of the reference's ``models/VisMVSNet/preproc.py`` that is on the hot path
(preproc.py:63-92, called at model_cas.py:177); the cv2 augmentation helpers there are unused by inference."""
import torch


def scale_camera(cam: torch.Tensor, scale=1):
    """Focal lengths and principal point of a [..,2,4,4] cam array times ``scale`` (a float or an (sx, sy) tuple)."""
    sx, sy = scale if isinstance(scale, tuple) else (scale, scale)
    out = cam.clone()
    out[..., 1, 0, 0] = cam[..., 1, 0, 0] * sx
    out[..., 1, 1, 1] = cam[..., 1, 1, 1] * sy
    out[..., 1, 0, 2] = cam[..., 1, 0, 2] * sx
    out[..., 1, 1, 2] = cam[..., 1, 1, 2] * sy
    return out
