"""maskrcnn-benchmark style names used by the SGG_emb model (lib/model/roi_layers): ``ROIPool``, ``ROIAlign``, ``nms``.

``ROIAlign(output_size, scale, sampling_ratio)`` is the sub-bin sampling variant bound to ``model._C`` in the reference
(roi_layers/roi_align.py:20); that source is absent from the reference tree, so the published maskrcnn-benchmark
algorithm is what ``i2v_roi_align_sampled_*`` implements (parity unpinned)."""
from torch import nn

from i2vsgg_amd import ops
from ..nms.nms_wrapper import nms  # noqa: F401


class ROIPool(nn.Module):
    def __init__(self, output_size, spatial_scale, out_nchw=True):
        super().__init__()
        self.output_size = tuple(output_size) if isinstance(output_size, (tuple, list)) else (output_size,) * 2
        self.spatial_scale = spatial_scale
        self.out_nchw = out_nchw     # NCHW: the flatten order vrd.fc6 expects (resnet_SGG_emb.py:146)

    def forward(self, input, rois):
        if isinstance(input, ops.PackedMaps):       # a captured step: the maps' extent is read on the device
            return ops.roi_pool_packed(input, rois, self.output_size[0], self.output_size[1], self.spatial_scale,
                                       out_nchw=self.out_nchw)
        return ops.roi_pool(input, rois, self.output_size[0], self.output_size[1], self.spatial_scale,
                            out_nchw=self.out_nchw)

    def __repr__(self):
        return "ROIPool(output_size=%s, spatial_scale=%s)" % (self.output_size, self.spatial_scale)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, out_nchw=False):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio
        self.out_nchw = out_nchw     # channels-last by default: what layer4's convolutions consume

    def forward(self, input, rois):
        h, w = (self.output_size if isinstance(self.output_size, (tuple, list)) else (self.output_size,) * 2)
        return ops.roi_align_sampled(input, rois, h, w, self.spatial_scale, self.sampling_ratio, out_nchw=self.out_nchw)

    def __repr__(self):
        return "ROIAlign(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (self.output_size, self.spatial_scale,
                                                                               self.sampling_ratio)
