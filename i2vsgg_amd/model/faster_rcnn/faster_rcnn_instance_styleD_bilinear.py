"""``_fasterRCNN`` of the instance_styleD stage (faster_rcnn/faster_rcnn_instance_styleD_bilinear.py:24-211).

forward(im_data, im_info, gt_boxes, num_boxes, target=False, eta=1.0, eta_style=1.0)
  source -> (rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox,
             rois_label, d_instance, d_style)
  target -> (d_instance, d_style)
Every dense stage runs on the HIP kernels (implicit-GEMM convs with fused BN/ReLU/residual, device
proposal layer, fused RoIAlignAvg, streaming bilinear pooling); torch supplies autograd, the tiny
loss arithmetic and the optimizer plumbing."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..roi_align.modules.roi_align import RoIAlignAvg
from ..roi_pooling.modules.roi_pool import _RoIPooling
from ..rpn.proposal_target_layer_cascade import _ProposalTargetLayer
from ..rpn.rpn import _RPN
from ..utils.config import cfg
from ..utils.net_utils import _smooth_l1_loss


class _fasterRCNN(nn.Module):
    def __init__(self, classes, class_agnostic, ic, gc):
        super().__init__()
        self.classes = classes
        self.n_classes = len(classes)
        self.class_agnostic = class_agnostic
        self.RCNN_loss_cls = 0
        self.RCNN_loss_bbox = 0
        self.ic, self.gc = ic, gc
        self.RCNN_rpn = _RPN(self.dout_base_model)
        self.RCNN_proposal_target = _ProposalTargetLayer(self.n_classes)
        self.RCNN_roi_pool = _RoIPooling(cfg.POOLING_SIZE, cfg.POOLING_SIZE, 1.0 / 16.0)
        self.RCNN_roi_align = RoIAlignAvg(cfg.POOLING_SIZE, cfg.POOLING_SIZE, 1.0 / 16.0)

    def forward(self, im_data, im_info, gt_boxes, num_boxes, target=False, eta=1.0, eta_style=1.0):
        base_feat, base_feat1 = self.extract_feature(im_data)
        return self.forward_features(base_feat, base_feat1, im_info, gt_boxes, num_boxes, target, eta, eta_style)

    def forward_features(self, base_feat, base_feat1, im_info, gt_boxes, num_boxes, target=False, eta=1.0, eta_style=1.0):
        """Everything of ``forward`` behind ``extract_feature`` (:62-182).  A training step that runs the backbone ONCE
        over the source and target frames of a D+G step (frozen BN: no cross-frame statistics, so one 8-frame pass equals
        two 4-frame passes) calls this twice, on the two halves of the feature maps (train.InstanceStyleDStep)."""
        batch_size = base_feat.size(0)
        im_info, gt_boxes, num_boxes = im_info.data, gt_boxes.data, num_boxes.data
        if self.gc:
            d_style, _ = self.netD_style(base_feat1, eta_style)
            if not target:
                _, feat_image = self.netD_style(base_feat1.detach(), eta_style)
        else:
            d_style = self.netD_style(base_feat1, eta_style)

        rois, rpn_loss_cls, rpn_loss_bbox = self.RCNN_rpn(base_feat, im_info, gt_boxes, num_boxes, target)
        if self.training and not target:
            rois, rois_label, rois_target, rois_inside_ws, rois_outside_ws = \
                self.RCNN_proposal_target(rois, gt_boxes, num_boxes)
            rois_label = rois_label.view(-1).long()
            rois_target = rois_target.view(-1, rois_target.size(2))
            rois_inside_ws = rois_inside_ws.view(-1, rois_inside_ws.size(2))
            rois_outside_ws = rois_outside_ws.view(-1, rois_outside_ws.size(2))
        else:
            rois_label = rois_target = rois_inside_ws = rois_outside_ws = None
            rpn_loss_cls = rpn_loss_bbox = 0

        if cfg.POOLING_MODE == "align":
            pooled_feat = self.RCNN_roi_align(base_feat, rois.view(-1, 5))
        elif cfg.POOLING_MODE == "pool":
            pooled_feat = self.RCNN_roi_pool(base_feat, rois.view(-1, 5))
        else:
            raise ValueError("POOLING_MODE %r: 'crop' is dead code in the reference and not provided" % cfg.POOLING_MODE)

        if self.ic:
            d_instance, _ = self.netD_pixel(pooled_feat, eta)
            if not target:
                _, feat_instance = self.netD_pixel(pooled_feat.detach(), eta)
        else:
            d_instance = self.netD_pixel(pooled_feat, eta)
        if target:
            return d_instance, d_style

        pooled_feat = self._head_to_tail(pooled_feat)
        if self.gc:
            n_prop = pooled_feat.size(0) // batch_size
            ctx = feat_image.unsqueeze(1).repeat(1, n_prop, 1).view(-1, feat_image.size(1))
            pooled_feat = torch.cat((ctx, pooled_feat), 1)
        if self.ic:
            pooled_feat = torch.cat((feat_instance.view(feat_instance.size(0), -1), pooled_feat), 1)

        bbox_pred = self.RCNN_bbox_pred(pooled_feat)
        if self.training and not self.class_agnostic:
            view = bbox_pred.view(bbox_pred.size(0), bbox_pred.size(1) // 4, 4)
            bbox_pred = torch.gather(view, 1, rois_label.view(-1, 1, 1).expand(-1, 1, 4)).squeeze(1)
        cls_score = self.RCNN_cls_score(pooled_feat)
        cls_prob = F.softmax(cls_score, 1)
        RCNN_loss_cls = RCNN_loss_bbox = 0
        if self.training:
            RCNN_loss_cls = F.cross_entropy(cls_score, rois_label)
            RCNN_loss_bbox = _smooth_l1_loss(bbox_pred, rois_target, rois_inside_ws, rois_outside_ws)
        cls_prob = cls_prob.view(batch_size, rois.size(1), -1)
        bbox_pred = bbox_pred.view(batch_size, rois.size(1), -1)
        if batch_size == 1 or not self.training:
            return (rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, RCNN_loss_bbox,
                    rois_label, d_instance, d_style)
        return (rois, cls_prob, bbox_pred, rpn_loss_cls.view(-1), rpn_loss_bbox.view(-1), RCNN_loss_cls.view(-1),
                RCNN_loss_bbox.view(-1), rois_label, d_instance, d_style)

    @torch.no_grad()
    def forward_detect(self, im_data, im_info):
        """What the test loop reads of an eval forward (test_net_instance_styleD_bilinear.py:140-149: rois, cls_prob,
        bbox_pred): the eval branch of ``forward`` without the two discriminator outputs nobody reads there -- netD_style /
        netD_pixel run only where their context vectors feed the classifier (gc / ic)."""
        if self.training:
            raise RuntimeError("forward_detect is the evaluation path: call .eval() first")
        if self.gc:
            base_feat, base_feat1 = self.extract_feature(im_data)
        else:
            base_feat = self.RCNN_base(im_data)
        batch_size = base_feat.size(0)
        rois, _, _ = self.RCNN_rpn(base_feat, im_info, None, None)
        if cfg.POOLING_MODE == "align":
            pooled_feat = self.RCNN_roi_align(base_feat, rois.view(-1, 5))
        elif cfg.POOLING_MODE == "pool":
            pooled_feat = self.RCNN_roi_pool(base_feat, rois.view(-1, 5))
        else:
            raise ValueError("POOLING_MODE %r: 'crop' is dead code in the reference and not provided" % cfg.POOLING_MODE)
        feat = self._head_to_tail(pooled_feat)
        if self.gc:
            _, feat_image = self.netD_style(base_feat1, 1.0)
            n_prop = feat.size(0) // batch_size
            feat = torch.cat((feat_image.unsqueeze(1).repeat(1, n_prop, 1).view(-1, feat_image.size(1)), feat), 1)
        if self.ic:
            _, feat_instance = self.netD_pixel(pooled_feat, 1.0)
            feat = torch.cat((feat_instance.view(feat_instance.size(0), -1), feat), 1)
        bbox_pred = self.RCNN_bbox_pred(feat)
        cls_prob = F.softmax(self.RCNN_cls_score(feat), 1)
        return rois, cls_prob.view(batch_size, rois.size(1), -1), bbox_pred.view(batch_size, rois.size(1), -1)

    def _init_weights(self):
        def normal_init(m, mean, std):
            m.weight.data.normal_(mean, std)
            if m.bias is not None:
                m.bias.data.zero_()
        normal_init(self.RCNN_rpn.RPN_Conv, 0, 0.01)
        normal_init(self.RCNN_rpn.RPN_cls_score, 0, 0.01)
        normal_init(self.RCNN_rpn.RPN_bbox_pred, 0, 0.01)
        normal_init(self.RCNN_cls_score, 0, 0.01)
        normal_init(self.RCNN_bbox_pred, 0, 0.001)

    def create_architecture(self):
        self._init_modules()
        self._init_weights()
