"""``_fasterRCNN`` of the SGG_emb stage (faster_rcnn/faster_rcnn_SGG_emb.py:32-379), pre_det task.

forward(im_data, im_info, gt_boxes, num_boxes, im_path, target=False)
  training -> BCE-with-logits loss of the relation head on the annotated (subject, object) pairs.
The reference moves the backbone output to the host, builds pairs / union boxes / masks in Python
loops and uploads the map again, one frame per step whatever ``--bs`` says.  Here the feature map never
leaves the GPU, ``im_path`` may be a list (one entry per frame; loss = mean over frames, i.e. the
reference run once per frame and averaged), pair tables are built with vectorised numpy (float64, the
reference's arithmetic) and the dual masks are rasterised on the device from their integer bounds."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..roi_layers import ROIAlign, ROIPool
from ..rpn.proposal_target_layer_cascade import _ProposalTargetLayer
from ..rpn.rpn import _RPN
from ..utils.config import cfg


def build_pair_tables(anno, im_scale, ih, iw, n_rel, margin=10):
    """forward_predicate (:170-245) for one frame: unique (s,o) pairs in first-seen order, multi-hot
    labels, union boxes (+margin, clipped to (iw, ih)) and the integer bounds of the 32x32 dual masks."""
    gt = np.array(anno["boxes"], dtype=np.float64).reshape(-1, 4) * im_scale
    index, ixs, ixo, prds = {}, [], [], []
    for s, o, r in anno["rels"]:
        key = (s, o)
        if key not in index:
            index[key] = len(ixs)
            ixs.append(s)
            ixo.append(o)
            prds.append([r])
        else:
            prds[index[key]].append(r)
    ixs, ixo = np.asarray(ixs, np.int64), np.asarray(ixo, np.int64)
    n = ixs.size
    labels = np.zeros((n, n_rel), np.float32)
    for i, rr in enumerate(prds):
        labels[i, rr] = 1
    sb, ob = gt[ixs], gt[ixo]
    union = np.stack([np.maximum(0, np.minimum(sb[:, 0], ob[:, 0]) - margin),
                      np.maximum(0, np.minimum(sb[:, 1], ob[:, 1]) - margin),
                      np.minimum(iw, np.maximum(sb[:, 2], ob[:, 2]) + margin),
                      np.minimum(ih, np.maximum(sb[:, 3], ob[:, 3]) + margin)], 1)

    def bounds(bb):          # _getDualMask (resnet_SGG_emb.py:246-256)
        rh, rw = 32.0 / ih, 32.0 / iw
        return np.stack([np.maximum(0, np.floor(bb[:, 0] * rw)), np.maximum(0, np.floor(bb[:, 1] * rh)),
                         np.minimum(32, np.ceil(bb[:, 2] * rw)), np.minimum(32, np.ceil(bb[:, 3] * rh))], 1)

    return gt, union, np.stack([bounds(sb), bounds(ob)], 1).astype(np.int32), labels, ixs, ixo


def build_eval_pair_tables(boxes_scaled, ih, iw, margin=10):
    """Eval branch of forward_relation (:597-652): ALL ordered pairs i != j of the frame's boxes (i-major), their union
    boxes (+margin, clipped to (iw, ih)) and the integer bounds of the 32x32 dual masks.  float64 like the reference."""
    b = np.asarray(boxes_scaled, np.float64).reshape(-1, 4)
    n = b.shape[0]
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    keep = ii != jj
    ixs, ixo = ii[keep].astype(np.int64), jj[keep].astype(np.int64)
    sb, ob = b[ixs], b[ixo]
    union = np.stack([np.maximum(0, np.minimum(sb[:, 0], ob[:, 0]) - margin),
                      np.maximum(0, np.minimum(sb[:, 1], ob[:, 1]) - margin),
                      np.minimum(iw, np.maximum(sb[:, 2], ob[:, 2]) + margin),
                      np.minimum(ih, np.maximum(sb[:, 3], ob[:, 3]) + margin)], 1)

    def bounds(bb):          # _getDualMask (resnet_SGG_emb.py:246-256)
        rh, rw = 32.0 / ih, 32.0 / iw
        return np.stack([np.maximum(0, np.floor(bb[:, 0] * rw)), np.maximum(0, np.floor(bb[:, 1] * rh)),
                         np.minimum(32, np.ceil(bb[:, 2] * rw)), np.minimum(32, np.ceil(bb[:, 3] * rh))], 1)

    return union, np.stack([bounds(sb), bounds(ob)], 1).astype(np.int32), ixs, ixo


def rasterize_masks(bounds, device):
    """(n,2,4) int [x1,y1,x2,y2) -> (n,2,32,32) float masks on the device."""
    b = torch.as_tensor(bounds, device=device).view(-1, 2, 4, 1, 1)
    ys = torch.arange(32, device=device).view(1, 1, 32, 1)
    xs = torch.arange(32, device=device).view(1, 1, 1, 32)
    return ((xs >= b[:, :, 0]) & (xs < b[:, :, 2]) & (ys >= b[:, :, 1]) & (ys < b[:, :, 3])).float()


class _fasterRCNN(nn.Module):
    def __init__(self, classes, args):
        super().__init__()
        self.classes = classes
        self.n_classes = len(classes)
        self.args = args
        self.RCNN_rpn = _RPN(self.dout_base_model)       # constructed, unused by forward (as in the reference)
        self.RCNN_proposal_target = _ProposalTargetLayer(self.n_classes)
        self.RCNN_roi_pool = ROIPool((cfg.POOLING_SIZE, cfg.POOLING_SIZE), 1.0 / 16.0, out_nchw=False)    # :46
        self.RCNN_roi_align = ROIAlign((cfg.POOLING_SIZE, cfg.POOLING_SIZE), 1.0 / 16.0, 0)               # :47

    def forward(self, im_data, im_info, gt_boxes, num_boxes, im_path, target=False):
        with torch.no_grad():                            # base_feat.detach() of the reference (:148)
            base_feat = self.RCNN_base(im_data)
        task = getattr(self.args, "vrd_task", "pre_det")
        if task != "pre_det":
            raise NotImplementedError("vrd_task=%r: only pre_det is live in the reference (SURVEY.md A10)" % task)
        if not self.training:
            # eval forward_predicate is broken in the reference; the live eval path is the
            # relation branch on the target annotations (forward_relation :583-697)
            return self.forward_relation_eval(base_feat, im_info, im_path if isinstance(im_path, str) else im_path[0])
        return self.forward_predicate(base_feat, im_info, im_path)

    @torch.no_grad()
    def forward_relation_eval(self, fmap, im_info, im_path):
        """Eval branch of forward_relation (:583-697) for ONE frame: the annotated boxes of ``target_gt_rels[im_path]``
        (confidence 1), every ordered pair, union boxes, dual masks, the subject/object prior rows, the relation head
        in eval mode (softmax over predicates) -> the reference's ``vrd_data`` dict, with ``rel_score`` / ``pre_feat``
        left on the device.  The feature map never visits the host (the reference copies it down and up, :148)."""
        anno = self.vrd.target_gt_rels[im_path]
        info = im_info.detach().cpu().numpy().reshape(-1, 3)
        ih, iw, sc = float(info[0][0]), float(info[0][1]), float(info[0][2])
        detected = anno["boxes"]
        classes = list(anno["box_classes"])
        scores = [1 for _ in classes]
        if len(detected) == 0:
            return {"bboxes": [], "classes": [], "scores": []}
        if len(detected) == 1:
            return {"bboxes": detected, "classes": classes, "scores": scores}
        boxes = np.array(detected, np.float64).reshape(-1, 4) * sc
        union, bnd, ixs, ixo = build_eval_pair_tables(boxes, ih, iw)
        dev = fmap.device
        cls = np.asarray(classes, np.int64)
        n_rel = self.vrd.n_rel
        so_prior = self.vrd._so_prior
        rel_so_prior = np.asarray(so_prior)[cls[ixs] - 1, cls[ixo] - 1] if so_prior is not None else np.zeros((ixs.size, n_rel))
        b5 = np.zeros((boxes.shape[0], 5), np.float32)
        b5[:, 1:] = boxes
        r5 = np.zeros((union.shape[0], 5), np.float32)
        r5[:, 1:] = union
        was_training = self.vrd.training
        self.vrd.eval()
        try:
            rel_score, pre_feat = self.vrd.forward_device(fmap[:1], torch.from_numpy(b5).to(dev), torch.from_numpy(r5).to(dev),
                                                          rasterize_masks(bnd, dev), torch.from_numpy(ixs).to(dev),
                                                          torch.from_numpy(ixo).to(dev))
        finally:
            self.vrd.train(was_training)
        return {"ixs": ixs, "ixo": ixo, "bboxes": detected, "classes": classes, "scores": scores, "rel_score": rel_score,
                "pre_feat": pre_feat, "rel_so_prior": rel_so_prior}

    def _rois_of(self, bboxes, device):
        b = bboxes.detach().float().cpu().numpy() if torch.is_tensor(bboxes) else np.asarray(bboxes, np.float32)
        b = b.reshape(-1, 4)
        return torch.from_numpy(np.hstack((np.zeros((b.shape[0], 1), np.float32), b.astype(np.float32)))).to(device)

    @torch.no_grad()
    def _extract_feature(self, base_feat, bboxes):
        """:381-392: ROIAlign (sampling grid, ``roi_layers.ROIAlign``) of frame 0 at ``bboxes`` (n,4, network-input
        pixels) -> layer4 -> spatial mean; returns the (n,2048) features as a numpy array like the reference."""
        if not torch.is_tensor(base_feat):
            base_feat = torch.from_numpy(np.asarray(base_feat, np.float32)).to(self.RCNN_cls_score.weight.device)
        pooled = self.RCNN_roi_align(base_feat, self._rois_of(bboxes, base_feat.device))
        return self._head_to_tail(pooled).detach().cpu().numpy()

    @torch.no_grad()
    def classify_boxes(self, base_feat, bboxes):
        """Box classification of the eval branch of forward_predicate (:278-291): ROIAlign -> ``_head_to_tail`` ->
        ``RCNN_cls_score`` -> softmax with the background column zeroed -> (classes, confs) per box."""
        pooled = self.RCNN_roi_align(base_feat, self._rois_of(bboxes, base_feat.device))
        prob = F.softmax(self.RCNN_cls_score(self._head_to_tail(pooled)), 1)
        prob[:, 0] = 0.0
        conf, cls = prob.max(1)
        return cls.cpu().numpy(), conf.cpu().numpy()

    def forward_predicate(self, fmap, im_info, im_path):
        paths = [im_path] if isinstance(im_path, str) else list(im_path)
        if len(paths) == 1 and fmap.size(0) > 1:
            fmap = fmap[:1]                              # the reference uses frame 0 only (:170,:207-208)
        info = im_info.detach().cpu().numpy()
        dev = fmap.device
        boxes, rel_boxes, bounds, labels, ixs, ixo, counts = [], [], [], [], [], [], []
        off = 0
        for f, path in enumerate(paths):
            anno = self.vrd.source_gt_rels[path]
            if len(anno["rels"]) < 1:
                continue
            ih, iw, sc = float(info[f][0]), float(info[f][1]), float(info[f][2])
            gt, union, bnd, lab, s, o = build_pair_tables(anno, sc, ih, iw, self.vrd.n_rel)
            b5 = np.zeros((gt.shape[0], 5), np.float32)
            b5[:, 0], b5[:, 1:] = f, gt
            r5 = np.zeros((union.shape[0], 5), np.float32)
            r5[:, 0], r5[:, 1:] = f, union
            boxes.append(b5); rel_boxes.append(r5); bounds.append(bnd); labels.append(lab)
            ixs.append(s + off); ixo.append(o + off); counts.append(lab.shape[0])
            off += gt.shape[0]
        if not counts:
            return {"boxes": [], "classes": [], "confs": []}
        t = lambda a, dt=torch.float32: torch.from_numpy(np.concatenate(a)).to(dev, dt)
        if self.vrd.spatial_type == 1:
            # the 8-d relative-location feature (resnet_SGG_emb.py:258-264).  The reference's loop never builds it (the call is
            # commented out at :229 and its ``type=bool`` flag cannot select the branch); here the flag works
            b_all, s_all, o_all = np.concatenate(boxes)[:, 1:], np.concatenate(ixs), np.concatenate(ixo)
            spatial = torch.from_numpy(np.stack([self.vrd._getRelativeLoc(b_all[i], b_all[j]) for i, j in zip(s_all, o_all)])
                                       .astype(np.float32)).to(dev)
        else:
            spatial = rasterize_masks(np.concatenate(bounds), dev)
        score, _ = self.vrd.forward_device(fmap, t(boxes), t(rel_boxes), spatial, t(ixs, torch.long), t(ixo, torch.long))
        target = t(labels)
        # mean over frames of the per-frame BCE mean (== reference run per frame, averaged)
        w = torch.cat([torch.full((c,), 1.0 / (c * len(counts))) for c in counts]).to(dev)
        from i2vsgg_amd import ops
        return ops.bce_rows(score, target, w)

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
