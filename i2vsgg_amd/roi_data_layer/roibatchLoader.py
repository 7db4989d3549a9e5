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
    """``device_prep=True`` (not in the reference): the image work of the item -- BGR swap, flip, mean
    subtraction, the resize to the 600-px scale and the placement into the padded batch canvas -- is left to the GPU
    (``i2v_image_prep``).  An item is then (uint8 HWC frame as decoded, meta = [flipped, canvas_h, canvas_w,
    scale], gt_boxes, num_boxes[, path]) with the SAME gt_boxes / im_info contract as the host form (im_info = canvas size +
    scale); use ``collate_device_prep`` as the DataLoader's collate_fn (frames of one minibatch differ in native size) and
    ``Step.stage_batch_u8``.  A quarter of the float blob's bytes cross PCIe and the host resize (~50 ms per frame and core:
    20+ cores to feed one GPU at 430 frames/s) leaves the loader.  Minibatches whose target ratio is exactly 1 (the
    reference crops those to a square, :182-190) come back with meta[1] = 0: stage them through the host form.  In test mode
    (``training=False``) an item is the frame alone, (uint8 frame, meta, the [1,1,1,1,1] placeholder, 0, path), its canvas the
    resized frame: ``eval.DetectStep.stage_u8`` / ``eval.RelationStep.stage_u8``."""

    def __init__(self, roidb, ratio_list, ratio_index, batch_size, num_classes, training=True, normalize=None,
                 seg_return=False, path_return=False, device_prep=False):
        self._roidb, self._num_classes = roidb, num_classes
        self.max_num_box = cfg.MAX_NUM_GT_BOXES
        self.training, self.normalize = training, normalize
        self.ratio_list, self.ratio_index, self.batch_size = ratio_list, ratio_index, batch_size
        self.data_size = len(ratio_list)
        self.seg_return, self.path_return = seg_return, path_return
        self.device_prep = bool(device_prep)
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

    def _getitem_device_prep(self, index, idx):
        from .minibatch import _gt_blob, _read_image
        import numpy.random as npr
        e = self._roidb[idx]
        scale_ind = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0]          # the draw of get_minibatch (minibatch.py:26)
        target = cfg.TRAIN.SCALES[scale_ind]
        im = np.ascontiguousarray(_read_image(e))
        if im.ndim == 2:
            im = np.repeat(im[:, :, None], 3, 2)
        H0, W0 = im.shape[:2]
        f = float(target) / float(min(H0, W0))
        ho, wo = int(np.rint(H0 * f)), int(np.rint(W0 * f))                        # cv2's dsize (i2v_image_prep_size)
        if not self.training:              # test mode (:70-76 of the host form): the frame alone, unpadded; canvas = the resized frame
            meta = torch.tensor([0.0, ho, wo, f, target], dtype=torch.float64)
            return (torch.from_numpy(im.astype(np.uint8, copy=False)), meta, torch.FloatTensor([1, 1, 1, 1, 1]), 0, e["image"])
        gt_np = _gt_blob(e, f)
        np.random.shuffle(gt_np)
        gt = torch.from_numpy(gt_np)
        ratio = float(self.ratio_list_batch[index])
        if e["need_crop"]:
            return torch.tensor([ho, wo, f], dtype=torch.float32)                  # the reference's bare im_info (:89-90)
        if ratio < 1:
            ch, cw = int(np.ceil(wo / ratio)), wo
        elif ratio > 1:
            ch, cw = ho, int(np.ceil(ho * ratio))
        else:
            ch, cw = 0, 0                                                          # square trim: host form only
        meta = torch.tensor([float(bool(e["flipped"])), ch, cw, f, target], dtype=torch.float64)
        keep = ((gt[:, 0] != gt[:, 2]) & (gt[:, 1] != gt[:, 3])).nonzero().view(-1)
        pad = torch.zeros(self.max_num_box, gt.size(1))
        n = 0
        if keep.numel():
            g = gt[keep]
            n = min(g.size(0), self.max_num_box)
            pad[:n] = g[:n]
        item = (torch.from_numpy(im.astype(np.uint8, copy=False)), meta, pad, n)
        return item + (e["image"],) if self.path_return else item

    def __getitem__(self, index):
        idx = int(self.ratio_index[index]) if self.training else index
        if self.device_prep:
            return self._getitem_device_prep(index, idx)
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


def collate_device_prep(items):
    """collate_fn for ``roibatchLoader(device_prep=True)``: frames stay a list (native sizes differ inside a minibatch), the
    rest is stacked as the default collate does -> (frames [n x (H,W,3) uint8], meta (n,5) float64, gt_boxes (n,MAX,5),
    num_boxes (n,)[, paths]).  A minibatch holding a bare im_info item (``need_crop``) collapses to that tensor, as with the
    host form (the loops skip it)."""
    for it in items:
        if torch.is_tensor(it):
            return it
    out = [[it[0] for it in items], torch.stack([it[1] for it in items]), torch.stack([it[2] for it in items]),
           torch.tensor([it[3] for it in items])]
    if len(items[0]) > 4:
        out.append([it[4] for it in items])
    return out
