"""``_RPN`` (rpn/rpn.py:17-110): RPN conv head on the MFMA implicit-GEMM kernels, proposal layer
as one device pass, anchor targets + losses for source-domain training."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from i2vsgg_amd import ops
from ..faster_rcnn.layers import ConvParams
from ..utils.config import cfg
from ..utils.net_utils import _smooth_l1_loss
from .anchor_target_layer import _AnchorTargetLayer
from .proposal_layer import _ProposalLayer


class _RPN(nn.Module):
    def __init__(self, din):
        super().__init__()
        self.din = din
        self.anchor_scales = cfg.ANCHOR_SCALES
        self.anchor_ratios = cfg.ANCHOR_RATIOS
        self.feat_stride = cfg.FEAT_STRIDE[0]
        n_anchor = len(self.anchor_scales) * len(self.anchor_ratios)
        self.nc_score_out = n_anchor * 2
        self.nc_bbox_out = n_anchor * 4
        self.RPN_Conv = ConvParams(din, 512, 3, bias=True)
        self.RPN_cls_score = ConvParams(512, self.nc_score_out, 1, bias=True)
        self.RPN_bbox_pred = ConvParams(512, self.nc_bbox_out, 1, bias=True)
        self.RPN_proposal = _ProposalLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios)
        self.RPN_anchor_target = _AnchorTargetLayer(self.feat_stride, self.anchor_scales, self.anchor_ratios)
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0

    @staticmethod
    def reshape(x, d):
        s = x.size()
        return x.view(s[0], int(d), int(float(s[1] * s[2]) / float(d)), s[3])

    def head(self, base_feat):
        x = ops.conv2d(base_feat, self.RPN_Conv.weight, None, self.RPN_Conv.bias, None, 1, 1, relu=True)
        cls = ops.conv2d(x, self.RPN_cls_score.weight, None, self.RPN_cls_score.bias)
        box = ops.conv2d(x, self.RPN_bbox_pred.weight, None, self.RPN_bbox_pred.bias)
        return cls, box

    def forward(self, base_feat, im_info, gt_boxes, num_boxes, target=False):
        B = base_feat.size(0)
        cls, box = self.head(base_feat)
        cfg_key = "TRAIN" if self.training else "TEST"
        rois = self.RPN_proposal((cls.detach(), box.detach(), im_info, cfg_key, "logits"), target=target)
        self.rpn_loss_cls = 0
        self.rpn_loss_box = 0
        if self.training and not target:
            assert gt_boxes is not None
            H, W = cls.size(2), cls.size(3)
            A = self.nc_score_out // 2
            labels, tg, inw, outw = self.RPN_anchor_target.targets_yxa(H, W, gt_boxes, im_info)
            # (y,x,a)-ordered views of the NHWC maps: no permute copies
            c = cls.permute(0, 2, 3, 1)                                        # (B,H,W,2A), contiguous view
            pair = torch.stack((c[..., :A], c[..., A:]), -1).reshape(-1, 2)     # (B*HWA, 2) = (bg, fg)
            # rpn.py:89-96 selects the anchors with label != -1 and averages their cross entropy; ignore_index does the
            # same sum over the same anchors without a data-dependent shape (nonzero() synchronises with the host)
            self.rpn_loss_cls = F.cross_entropy(pair, labels.reshape(-1).long(), ignore_index=-1)
            pred = box.permute(0, 2, 3, 1).reshape(B, -1, 4)
            self.rpn_loss_box = _smooth_l1_loss(pred, tg, inw.unsqueeze(2), outw.unsqueeze(2), sigma=3, dim=[1, 2])
        return rois, self.rpn_loss_cls, self.rpn_loss_box
