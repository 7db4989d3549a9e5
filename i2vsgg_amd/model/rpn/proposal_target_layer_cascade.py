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
        self.device_sampling = False     # True: sample on the device from torch's generator (capturable; see _sample_device)
        self.sample_record = None        # parity instrumentation, as in _AnchorTargetLayer: {} -> (keep, nfg) of every call
        self.sample_replay = None        # such a dict -> used instead of drawing
        self._const = {}                 # device copies of the three constant vectors (an H2D copy cannot be captured)

    def _c(self, name, dev):
        key = (name, str(dev))
        if key not in self._const:
            self._const[key] = getattr(self, name).to(dev)
        return self._const[key]

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
        if self.sample_replay is not None:
            keep, nfg = self.sample_replay["keep"].to(dev), self.sample_replay["nfg"].to(dev)
            return self._emit(all_rois, gt_boxes, assign, keep, nfg, R)
        if self.device_sampling:
            keep, nfg = self._sample_device(max_ov, R, fg_per)
            if self.sample_record is not None:
                from .anchor_target_layer import record_sample
                record_sample(self.sample_record, "keep", keep)
                record_sample(self.sample_record, "nfg", nfg)
                record_sample(self.sample_record, "max_ov", max_ov)
            return self._emit(all_rois, gt_boxes, assign, keep, nfg, R)
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
        return self._emit(all_rois, gt_boxes, assign, keep, nfg, R)

    @staticmethod
    def _sample_device(max_ov, R, fg_per):
        """proposal_target_layer_cascade.py:140-182 without leaving the device.  Per image: n_fg = min(fg_per, #fg) foreground
        rois without replacement (a random permutation's head), the other R - n_fg slots from the background WITH
        replacement (floor(rand * #bg), as the reference does); only-fg / only-bg images fill all R slots with replacement
        from the class they have.  Fixed shapes, no host value.  (An image with neither raises in the reference; here its
        slots point at roi 0.)"""
        T = cfg.TRAIN
        B, N = max_ov.shape
        dev = max_ov.device
        fg = max_ov >= T.FG_THRESH
        bg = (max_ov < T.BG_THRESH_HI) & (max_ov >= T.BG_THRESH_LO)
        n_fg_all, n_bg_all = fg.sum(1, keepdim=True), bg.sum(1, keepdim=True)
        # compacted candidate lists: a random order of the fg rois first (keys in [0,1) for fg, 2 otherwise), index order for bg
        fg_list = torch.where(fg, torch.rand(B, N, device=dev), torch.full((B, N), 2.0, device=dev)).argsort(1)
        bg_list = torch.argsort((~bg).to(torch.int8), dim=1, stable=True)
        slot = torch.arange(R, device=dev).view(1, R)
        both = (n_fg_all > 0) & (n_bg_all > 0)
        nfg = torch.where(both, n_fg_all.clamp(max=fg_per), torch.where(n_fg_all > 0, torch.full_like(n_fg_all, R),
                                                                       torch.zeros_like(n_fg_all)))
        u = torch.rand(B, R, device=dev)
        # fg slots: permutation head when both classes exist, else with replacement
        fg_pick = torch.where(both, slot.expand(B, R), (u * n_fg_all).floor().long().clamp(max=N - 1))
        bg_pick = (u * n_bg_all).floor().long().clamp(min=0, max=N - 1)
        keep = torch.where(slot < nfg, torch.gather(fg_list, 1, fg_pick.clamp(max=N - 1)), torch.gather(bg_list, 1, bg_pick))
        return keep, nfg

    def _emit(self, all_rois, gt_boxes, assign, keep, nfg, R):
        dev = gt_boxes.device
        T = cfg.TRAIN
        B = gt_boxes.size(0)
        rois = torch.gather(all_rois, 1, keep.unsqueeze(2).expand(-1, -1, 5)).clone()
        rois[:, :, 0] = torch.arange(B, device=dev, dtype=rois.dtype).view(B, 1)
        gsel = torch.gather(assign.long(), 1, keep)
        gt_sel = torch.gather(gt_boxes, 1, gsel.unsqueeze(2).expand(-1, -1, 5))
        labels = torch.where(torch.arange(R, device=dev).view(1, R) >= nfg, torch.zeros_like(gt_sel[:, :, 4]),
                             gt_sel[:, :, 4])                                      # :196-197
        if rois.is_cuda:      # transform + normalisation in one kernel (gt_sel's rows are 5 wide: the box is its first four columns)
            norm = bool(T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED)
            tg = ops.bbox_transform(rois[:, :, 1:5], gt_sel, T.BBOX_NORMALIZE_MEANS if norm else None,
                                    T.BBOX_NORMALIZE_STDS if norm else None)
        else:
            tg = bbox_transform_batch(rois[:, :, 1:5], gt_sel[:, :, :4])
            if T.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
                tg = (tg - self._c("BBOX_NORMALIZE_MEANS", dev)) / self._c("BBOX_NORMALIZE_STDS", dev)
        fgmask = (labels > 0).unsqueeze(2).float()
        targets = tg * fgmask
        inw = fgmask * self._c("BBOX_INSIDE_WEIGHTS", dev).view(1, 1, 4)
        outw = (inw > 0).float()
        return rois, labels, targets, inw, outw
