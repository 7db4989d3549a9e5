"""``RoIAlign`` / ``RoIAlignAvg`` / ``RoIAlignMax`` with the reference signatures
(roi_align/modules/roi_align.py:6-42): ``Module(aligned_h, aligned_w, spatial_scale)(features, rois)``
-> (K, C, h, w); differentiable w.r.t. ``features`` only.  ``RoIAlignAvg`` is ONE fused kernel
(sample (h+1)x(w+1) points, 2x2 stride-1 mean in registers) instead of align + avg_pool2d."""
from torch.nn.functional import max_pool2d
from torch.nn.modules.module import Module

from i2vsgg_amd import ops


class _Base(Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale, out_nchw=False):
        super().__init__()
        self.aligned_height, self.aligned_width = int(aligned_height), int(aligned_width)
        self.spatial_scale = float(spatial_scale)
        self.out_nchw = out_nchw    # False: channels_last output for the HIP heads (same logical shape)


class RoIAlign(_Base):
    def forward(self, features, rois):
        return ops.roi_align(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale,
                             avg=False, out_nchw=self.out_nchw)


class RoIAlignAvg(_Base):
    def forward(self, features, rois):
        return ops.roi_align(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale,
                             avg=True, out_nchw=self.out_nchw)


class RoIAlignMax(_Base):
    def forward(self, features, rois):
        x = ops.roi_align(features, rois, self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale,
                          avg=False, out_nchw=self.out_nchw)
        return max_pool2d(x, kernel_size=2, stride=1)
