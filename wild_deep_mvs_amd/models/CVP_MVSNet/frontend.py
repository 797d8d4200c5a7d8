"""Drop-in for the reference's ``models/CVP_MVSNet/frontend.py``: common ``forward()`` -> CVP ``network``."""
import torch
from torch import nn

from ...graph import ReplayHooks, replayable
from .models.net import network


class Frontend(ReplayHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.model = network()

    @property
    def storage_dtype(self):
        return self.model.storage_dtype

    @storage_dtype.setter
    def storage_dtype(self, dt):
        self.model.storage_dtype = dt

    @property
    def train_storage_dtype(self):
        return self.model.train_storage_dtype

    @train_storage_dtype.setter
    def train_storage_dtype(self, dt):
        self.model.train_storage_dtype = dt

    @property
    def feature_engine(self):
        return self.model.feature_engine

    @feature_engine.setter
    def feature_engine(self, name):
        self.model.feature_engine = name

    @property
    def feature_engine_train(self):
        return self.model.feature_engine_train

    @feature_engine_train.setter
    def feature_engine_train(self, name):
        self.model.feature_engine_train = name

    def set_row_group(self, group):
        """Shard the image rows of the refinement levels over a torch.distributed group (None = no sharding); see
        ``network._refine_level_row_shard``."""
        self.model.row_group = group

    @replayable
    def forward(self, imgs, K, R, t, depth_min, depth_max, reference_frame=0, **kwargs):
        src_idx = [i for i in range(K.shape[1]) if i != reference_frame]
        if isinstance(imgs, torch.Tensor):
            ref_img, src_imgs = imgs[:, reference_frame], [imgs[:, i] for i in src_idx]
        else:
            ref_img, src_imgs = imgs[reference_frame], [imgs[i] for i in src_idx]
        b, n = ref_img.shape[0], len(src_imgs)
        # (0,0,0,1) built on the device: a host list / scalar assignment is a synchronous copy, which a hipGraph capture of the
        # forward does not permit
        row = (torch.arange(4, device=K.device) == 3).to(K.dtype)
        ref_ex = torch.cat((torch.cat((R[:, reference_frame], t[:, reference_frame]), dim=2), row.view(1, 1, 4).expand(b, 1, 4)), dim=1)
        # (views are picked with python slices, not index lists: an index list is a host tensor, i.e. a synchronous copy)
        pick = lambda x: torch.stack([x[:, i] for i in src_idx], dim=1)
        src_ex = torch.cat((torch.cat((pick(R), pick(t)), dim=3), row.view(1, 1, 1, 4).expand(b, n, 1, 4)), dim=2)
        out = self.model(ref_img, src_imgs, K[:, reference_frame], pick(K), ref_ex, src_ex,
                         depth_min[:, reference_frame], depth_max[:, reference_frame], **kwargs)
        return {"depth": out["depth_est_list"][0], "depth_est_list": out["depth_est_list"], "depth_pair_list": [],
                "photometric_confidence": out["prob_confidence"].unsqueeze(1)}
