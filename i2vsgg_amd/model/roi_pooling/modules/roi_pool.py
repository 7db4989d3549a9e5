"""``_RoIPooling(h, w, scale)(features, rois)`` (roi_pooling/modules/roi_pool.py): Caffe max ROI pooling."""
from torch.nn.modules.module import Module

from i2vsgg_amd import ops


class _RoIPooling(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale, out_nchw=False):
        super().__init__()
        self.pooled_height, self.pooled_width = int(pooled_height), int(pooled_width)
        self.spatial_scale = float(spatial_scale)
        self.out_nchw = out_nchw

    def forward(self, features, rois):
        return ops.roi_pool(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale,
                            out_nchw=self.out_nchw)
