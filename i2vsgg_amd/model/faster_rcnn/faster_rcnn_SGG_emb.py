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

    def forward(self, im_data, im_info, gt_boxes, num_boxes, im_path, target=False):
        with torch.no_grad():                            # base_feat.detach() of the reference (:148)
            base_feat = self.RCNN_base(im_data)
        task = getattr(self.args, "vrd_task", "pre_det")
        if task != "pre_det":
            raise NotImplementedError("vrd_task=%r: only pre_det is live in the reference (SURVEY.md A10)" % task)
        if not self.training:
            raise NotImplementedError("eval forward_predicate is broken in the reference (SURVEY.md A9, A11); "
                                      "relation scoring is listed as next (SURVEY.md 8f row f3)")
        return self.forward_predicate(base_feat, im_info, im_path)

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
        score, _ = self.vrd.forward_device(fmap, t(boxes), t(rel_boxes), rasterize_masks(np.concatenate(bounds), dev),
                                           t(ixs, torch.long), t(ixo, torch.long))
        target = t(labels)
        # mean over frames of the per-frame BCE mean (== reference run per frame, averaged)
        per = nn.functional.binary_cross_entropy_with_logits(score, target, reduction="none").mean(1)
        w = torch.cat([torch.full((c,), 1.0 / (c * len(counts))) for c in counts]).to(dev)
        return (per * w).sum()

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
