"""``_ProposalTargetLayer`` (rpn/proposal_target_layer_cascade.py:20-212): append GT to the
proposals, IoU on the device (HIP kernel), fg/bg sampling on the host with the reference's
``np.random`` call order (:158, :167, :174, :182), class-agnostic 4-d targets normalised by the
configured means/stds."""
import numpy as np
import torch
import torch.nn as nn

from i2vsgg_amd import ops
from ..utils.config import cfg
from .bbox_transform import bbox_transform_batch


class _ProposalTargetLayer(nn.Module):
    def __init__(self, nclasses):
        super().__init__()
        self._num_classes = nclasses
        self.BBOX_NORMALIZE_MEANS = torch.FloatTensor(cfg.TRAIN.BBOX_NORMALIZE_MEANS)
        self.BBOX_NORMALIZE_STDS = torch.FloatTensor(cfg.TRAIN.BBOX_NORMALIZE_STDS)
        self.BBOX_INSIDE_WEIGHTS = torch.FloatTensor(cfg.TRAIN.BBOX_INSIDE_WEIGHTS)

    def forward(self, all_rois, gt_boxes, num_boxes):
        dev = gt_boxes.device
        T = cfg.TRAIN
        B = gt_boxes.size(0)
        app = torch.zeros_like(gt_boxes)
        app[:, :, 1:5] = gt_boxes[:, :, :4]
        all_rois = torch.cat([all_rois, app], 1)                                  # :41-45
        R = int(T.BATCH_SIZE)
        fg_per = int(np.round(T.FG_FRACTION * R)) or 1
        _, max_ov, assign = ops.bbox_overlaps(all_rois, gt_boxes)
        mo = max_ov.cpu().numpy()                                                 # the one D2H (B x (P+G) floats)
        keep_all, nfg_all = [], []
        for i in range(B):
            fg = np.nonzero(mo[i] >= T.FG_THRESH)[0]
            bg = np.nonzero((mo[i] < T.BG_THRESH_HI) & (mo[i] >= T.BG_THRESH_LO))[0]
            if fg.size > 0 and bg.size > 0:
                nfg = min(fg_per, fg.size)
                fg = fg[np.random.permutation(fg.size)[:nfg]]
                bg = bg[np.floor(np.random.rand(R - nfg) * bg.size).astype(np.int64)]
            elif fg.size > 0:
                fg = fg[np.floor(np.random.rand(R) * fg.size).astype(np.int64)]
                nfg, bg = R, bg[:0]
            elif bg.size > 0:
                bg = bg[np.floor(np.random.rand(R) * bg.size).astype(np.int64)]
                nfg, fg = 0, fg[:0]
            else:
                raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
            keep_all.append(np.concatenate([fg, bg]))
            nfg_all.append(nfg)
        keep = torch.from_numpy(np.stack(keep_all)).to(dev)                        # (B,R)
        nfg = torch.tensor(nfg_all, device=dev).view(B, 1)
        rois = torch.gather(all_rois, 1, keep.unsqueeze(2).expand(-1, -1, 5)).clone()
        rois[:, :, 0] = torch.arange(B, device=dev, dtype=rois.dtype).view(B, 1)
        gsel = torch.gather(assign.long(), 1, keep)
        gt_sel = torch.gather(gt_boxes, 1, gsel.unsqueeze(2).expand(-1, -1, 5))
        labels = gt_sel[:, :, 4].clone()
        labels[torch.arange(R, device=dev).view(1, R) >= nfg] = 0                  # :196-197
        tg = bbox_transform_batch(rois[:, :, 1:5], gt_sel[:, :, :4])
        if T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
            tg = (tg - self.BBOX_NORMALIZE_MEANS.to(dev)) / self.BBOX_NORMALIZE_STDS.to(dev)
        fgmask = (labels > 0).unsqueeze(2).float()
        targets = tg * fgmask
        inw = fgmask * self.BBOX_INSIDE_WEIGHTS.to(dev).view(1, 1, 4)
        outw = (inw > 0).float()
        return rois, labels, targets, inw, outw
