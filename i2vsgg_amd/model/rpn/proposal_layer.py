"""``_ProposalLayer`` (rpn/proposal_layer.py:20-163) as ONE asynchronous device pass.

The reference builds anchors on the host every call, decodes with ~10 element-wise kernels, sorts,
then copies every image to the host for a numpy NMS.  Here anchors + decode + clip, the per-image
descending sort, the bitmask NMS with its suppression scan, the top-N selection and the zero padding
all stay on the GPU inside ``i2v_rpn_proposal``; nothing synchronises."""
import numpy as np
import torch
import torch.nn as nn

from i2vsgg_amd import ops
from ..utils.config import cfg
from .generate_anchors import generate_anchors


class _ProposalLayer(nn.Module):
    def __init__(self, feat_stride, scales, ratios):
        super().__init__()
        self._feat_stride = feat_stride
        base = generate_anchors(scales=np.array(scales), ratios=np.array(ratios))
        self.register_buffer("_anchors", torch.from_numpy(base).float(), persistent=False)
        self._num_anchors = base.shape[0]

    def forward(self, input, target=False, want_index=False):
        """input = (scores, bbox_deltas, im_info, cfg_key).  ``scores`` is either the (B,2A,H,W)
        probability map of the reference API or, with ``input[4] == 'logits'``, the raw class scores
        (the pairwise softmax of rpn.py:69-71 is then fused into the decode kernel)."""
        scores, deltas, im_info, cfg_key = input[:4]
        is_prob = not (len(input) > 4 and input[4] == "logits")
        c = cfg[cfg_key]
        post = c.RPN_POST_NMS_TOP_N_TARGET if target else c.RPN_POST_NMS_TOP_N       # :72-75 (reads cfg per call)
        if self._anchors.device != scores.device:
            self._anchors = self._anchors.to(scores.device)
        return ops.rpn_proposal(scores, deltas, im_info, self._anchors, self._feat_stride, c.RPN_PRE_NMS_TOP_N, post,
                                c.RPN_NMS_THRESH, is_prob=is_prob, want_index=want_index)
