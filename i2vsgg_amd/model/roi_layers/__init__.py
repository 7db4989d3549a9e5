"""maskrcnn-benchmark style names used by the SGG_emb model (lib/model/roi_layers): ``ROIPool`` and ``nms``.

``ROIAlign(output_size, scale, sampling_ratio)`` of the same package is only reached by the
reference's (broken) eval branches and its source (``model._C``) is absent from the
reference tree; it is listed as "next"  and raises here rather than guess."""
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
        return ops.roi_pool(input, rois, self.output_size[0], self.output_size[1], self.spatial_scale,
                            out_nchw=self.out_nchw)

    def __repr__(self):
        return "ROIPool(output_size=%s, spatial_scale=%s)" % (self.output_size, self.spatial_scale)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        raise NotImplementedError("roi_layers.ROIAlign (sub-bin sampling variant) is out of scope for this round: "
                                  "its reference source (model._C) is absent; see DESIGN.md 'Out of scope'")
