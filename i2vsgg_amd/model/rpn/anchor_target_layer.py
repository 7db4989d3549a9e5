"""``_AnchorTargetLayer`` (rpn/anchor_target_layer.py:30-193).

IoU of the inside anchors against the ground truth runs in a HIP kernel and the labelling is
device arithmetic; only the fg/bg SUBSAMPLING is done on the host, because the reference draws it
from the global ``np.random`` stream in a fixed call order (:131, :143) and that order is part of
the contract (cfg.RNG_SEED).  ``device_sampling = True`` (set by a step object that captures the step into a HIP graph)
draws the same subsample sizes from torch's device generator instead: no host round trip, a different (equally uniform)
random stream.  Quirks kept for parity: the inside test uses ``im_info[0]`` for every
image (:85-86) and the loss weights use the example count of the LAST image (:156-160)."""
import numpy as np
import torch
import torch.nn as nn

from i2vsgg_amd import ops
from ..utils.config import cfg
from .bbox_transform import bbox_transform_batch
from .generate_anchors import generate_anchors, shifted_anchors


def record_sample(rec, key, t):
    """Copy ``t`` into the static tensor ``rec[key]`` (made at first sight, outside any capture: a step's warm-up call)."""
    buf = rec.get(key)
    if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
        if torch.cuda.is_available() and t.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sample_record: no buffer for %r yet -- run one eager step before capturing" % key)
        rec[key] = t.detach().clone()
    else:
        buf.copy_(t)


class _AnchorTargetLayer(nn.Module):
    def __init__(self, feat_stride, scales, ratios):
        super().__init__()
        self._feat_stride = feat_stride
        self._base = generate_anchors(scales=np.array(scales), ratios=np.array(ratios))
        self._num_anchors = self._base.shape[0]
        self._allowed_border = 0
        self._cache = {}
        self.device_sampling = False     # True: subsample on the device (capturable), torch's generator
        # parity instrumentation: ``sample_record`` = {} -> the subsampled labels of every call are copied into it (static
        # tensors, so a captured step records too); ``sample_replay`` = such a dict -> its labels are used instead of drawing
        self.sample_record = None
        self.sample_replay = None
        self.image_size = None           # (h, w) of im_info[0] known on the host: skips the .tolist() synchronisation

    def _anchors_for(self, H, W, imh, imw, device):
        key = (H, W, imh, imw, str(device))
        if key not in self._cache:
            allanc = shifted_anchors(H, W, self._feat_stride, self._base)
            b = self._allowed_border
            inside = np.nonzero((allanc[:, 0] >= -b) & (allanc[:, 1] >= -b) &
                                (allanc[:, 2] < imw + b) & (allanc[:, 3] < imh + b))[0]
            self._cache[key] = (torch.from_numpy(allanc[inside]).to(device), torch.from_numpy(inside).to(device),
                                allanc.shape[0])
        return self._cache[key]

    def targets_yxa(self, H, W, gt_boxes, im_info):
        """labels (B,HWA), targets (B,HWA,4), inside-w, outside-w (B,HWA) in (y,x,a) order."""
        B = gt_boxes.size(0)
        dev = gt_boxes.device
        info0 = self.image_size if self.image_size is not None else im_info[0].tolist()
        anc, inside, total = self._anchors_for(H, W, int(info0[0]), int(info0[1]), dev)
        ov, max_ov, argmax = ops.bbox_overlaps(anc, gt_boxes, want_matrix=True)
        T = cfg.TRAIN
        neg, one, zero = torch.full_like(max_ov, -1.0), torch.ones_like(max_ov), torch.zeros_like(max_ov)
        labels = neg                                                         # torch.where throughout: no index_put
        if not T.RPN_CLOBBER_POSITIVES:                                      # with a boolean mask (it synchronises)
            labels = torch.where(max_ov < T.RPN_NEGATIVE_OVERLAP, zero, labels)
        gt_max = ov.max(1)[0]
        gt_max = torch.where(gt_max == 0, torch.full_like(gt_max, 1e-5), gt_max)
        labels = torch.where((ov == gt_max.unsqueeze(1)).sum(2) > 0, one, labels)
        labels = torch.where(max_ov >= T.RPN_POSITIVE_OVERLAP, one, labels)
        if T.RPN_CLOBBER_POSITIVES:
            labels = torch.where(max_ov < T.RPN_NEGATIVE_OVERLAP, zero, labels)
        num_fg = int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE)
        assert T.RPN_POSITIVE_WEIGHT < 0, "only the uniform weighting of the reference recipes is supported"
        if self.sample_replay is not None:
            labels = self.sample_replay["labels"].to(device=dev, dtype=labels.dtype)
            w = 1.0 / (labels[B - 1] >= 0).sum().float()
        elif self.device_sampling:
            labels = self._subsample_device(labels, num_fg, int(T.RPN_BATCHSIZE))
            w = 1.0 / (labels[B - 1] >= 0).sum().float()
        else:
            # --- subsampling on the host with the reference's np.random call order
            lab = labels.cpu().numpy()
            sum_fg, sum_bg = (lab == 1).sum(1), (lab == 0).sum(1)
            for i in range(B):
                if sum_fg[i] > num_fg:
                    fg = np.nonzero(lab[i] == 1)[0]
                    lab[i, fg[np.random.permutation(fg.size)[:fg.size - num_fg]]] = -1
                num_bg = T.RPN_BATCHSIZE - int((lab[i] == 1).sum())
                if sum_bg[i] > num_bg:
                    bg = np.nonzero(lab[i] == 0)[0]
                    lab[i, bg[np.random.permutation(bg.size)[:bg.size - num_bg]]] = -1
            w = 1.0 / float((lab[B - 1] >= 0).sum())
            labels = torch.from_numpy(lab).to(dev)
        if self.sample_record is not None:
            record_sample(self.sample_record, "labels", labels)
        gt_sel = torch.gather(gt_boxes[:, :, :4], 1, argmax.long().unsqueeze(2).expand(-1, -1, 4))
        tg = bbox_transform_batch(anc, gt_sel)
        inw = (labels == 1).float() * T.RPN_BBOX_INSIDE_WEIGHTS[0]
        outw = (labels >= 0).float() * w

        def unmap(x, fill):
            full = x.new_full((B, total) + tuple(x.shape[2:]), fill)
            full[:, inside] = x
            return full

        return unmap(labels, -1), unmap(tg, 0), unmap(inw, 0), unmap(outw, 0)

    @staticmethod
    def _subsample_device(labels, num_fg, batchsize):
        """anchor_target_layer.py:123-143 without leaving the device: keep at most ``num_fg`` foreground anchors and
        ``batchsize - #fg`` background anchors per image, chosen uniformly (rank of an i.i.d. uniform key among the
        candidates), the rest -> -1.  Fixed shapes, no host value: capturable."""
        B, N = labels.shape
        ar = torch.arange(N, device=labels.device).expand(B, N)

        def rank_among(mask):
            key = torch.where(mask, torch.rand_like(labels), torch.full_like(labels, 2.0))
            order = key.argsort(1)
            return torch.empty_like(order).scatter_(1, order, ar)

        fg = labels == 1
        labels = torch.where(fg & (rank_among(fg) >= num_fg), torch.full_like(labels, -1.0), labels)
        n_bg = batchsize - (labels == 1).sum(1, keepdim=True)
        bg = labels == 0
        return torch.where(bg & (rank_among(bg) >= n_bg), torch.full_like(labels, -1.0), labels)

    def forward(self, input):
        """Reference API: input = (rpn_cls_score, gt_boxes, im_info, num_boxes) ->
        [labels (B,1,A*H,W), targets, inside-w, outside-w (B,4A,H,W)]."""
        score, gt_boxes, im_info = input[0], input[1], input[2]
        H, W = score.size(2), score.size(3)
        B, A = gt_boxes.size(0), self._num_anchors
        L, T, IW, OW = self.targets_yxa(H, W, gt_boxes, im_info)
        L = L.view(B, H, W, A).permute(0, 3, 1, 2).contiguous().view(B, 1, A * H, W)
        T = T.view(B, H, W, 4 * A).permute(0, 3, 1, 2).contiguous()
        IW = IW.unsqueeze(2).expand(B, -1, 4).reshape(B, H, W, 4 * A).permute(0, 3, 1, 2).contiguous()
        OW = OW.unsqueeze(2).expand(B, -1, 4).reshape(B, H, W, 4 * A).permute(0, 3, 1, 2).contiguous()
        return [L, T, IW, OW]
