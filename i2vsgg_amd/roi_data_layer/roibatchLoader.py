"""``roibatchLoader`` Dataset (lib/roi_data_layer/roibatchLoader.py:22-221).

``loader[i]`` -> (data (3,H,W), im_info (3,), gt_boxes (MAX_NUM_GT_BOXES,5) zero padded, num_boxes[, path]).
Kept behaviours: images whose aspect ratio is outside [0.5, 2] return a BARE ``im_info`` (the whole
crop branch is dead code behind an early return, :89-90; the loops skip non-list items); images with no
boxes return ``(data, im_info)`` (:193-194); test mode returns gt_boxes = [1,1,1,1,1], num_boxes = 0 and
the path (:214-221)."""
import numpy as np
import torch
import torch.utils.data as data

from ..model.utils.config import cfg
from .minibatch import get_minibatch


class roibatchLoader(data.Dataset):
    def __init__(self, roidb, ratio_list, ratio_index, batch_size, num_classes, training=True, normalize=None,
                 seg_return=False, path_return=False):
        self._roidb, self._num_classes = roidb, num_classes
        self.max_num_box = cfg.MAX_NUM_GT_BOXES
        self.training, self.normalize = training, normalize
        self.ratio_list, self.ratio_index, self.batch_size = ratio_list, ratio_index, batch_size
        self.data_size = len(ratio_list)
        self.seg_return, self.path_return = seg_return, path_return
        # one target aspect ratio per batch so that its images pad to a common shape (:39-54)
        self.ratio_list_batch = torch.zeros(self.data_size)
        for i in range(int(np.ceil(len(ratio_index) / batch_size))):
            lo, hi = i * batch_size, min((i + 1) * batch_size - 1, self.data_size - 1)
            if ratio_list[hi] < 1:
                target = ratio_list[lo]
            elif ratio_list[lo] > 1:
                target = ratio_list[hi]
            else:
                target = 1.0
            self.ratio_list_batch[lo:hi + 1] = float(target)

    def __getitem__(self, index):
        idx = int(self.ratio_index[index]) if self.training else index
        blobs = get_minibatch([self._roidb[idx]], self._num_classes)
        img = torch.from_numpy(blobs["data"])                  # (1,H,W,3)
        im_info = torch.from_numpy(blobs["im_info"])
        H, W = img.size(1), img.size(2)
        if not self.training:
            return (img.permute(0, 3, 1, 2).contiguous().view(3, H, W), im_info.view(3),
                    torch.FloatTensor([1, 1, 1, 1, 1]), 0, blobs["path"])
        np.random.shuffle(blobs["gt_boxes"])
        gt = torch.from_numpy(blobs["gt_boxes"])
        ratio = float(self.ratio_list_batch[index])
        if self._roidb[idx]["need_crop"]:
            return im_info
        if ratio < 1:                                          # pad the height
            out = torch.zeros(int(np.ceil(W / ratio)), W, 3)
            out[:H] = img[0]
            im_info[0, 0] = out.size(0)
        elif ratio > 1:                                        # pad the width
            out = torch.zeros(H, int(np.ceil(H * ratio)), 3)
            out[:, :W] = img[0]
            im_info[0, 1] = out.size(1)
        else:                                                  # square trim
            t = min(H, W)
            out = img[0][:t, :t]
            gt[:, :4].clamp_(0, t)
            im_info[0, 0] = im_info[0, 1] = t
        out = out.permute(2, 0, 1).contiguous()
        im_info = im_info.view(3)
        if gt.shape[0] <= 0:
            return out, im_info
        keep = ((gt[:, 0] != gt[:, 2]) & (gt[:, 1] != gt[:, 3])).nonzero().view(-1)
        pad = torch.zeros(self.max_num_box, gt.size(1))
        n = 0
        if keep.numel():
            gt = gt[keep]
            n = min(gt.size(0), self.max_num_box)
            pad[:n] = gt[:n]
        if self.path_return:
            return out, im_info, pad, n, blobs["path"]
        return out, im_info, pad, n

    def __len__(self):
        return len(self._roidb)
