"""Box utilities with the reference names (rpn/bbox_transform.py).  IoU runs in a HIP kernel;
the element-wise transforms are torch expressions on device tensors (fp32, one rounding per op)."""
import torch

from i2vsgg_amd import ops


def _whc(b):
    w = b[..., 2] - b[..., 0] + 1.0
    h = b[..., 3] - b[..., 1] + 1.0
    return w, h, b[..., 0] + 0.5 * w, b[..., 1] + 0.5 * h


def bbox_transform_batch(ex_rois, gt_rois):
    """(N,4)|(B,N,4) vs (B,N,4) -> (B,N,4) regression targets (bbox_transform.py:36-75)."""
    if gt_rois.is_cuda and gt_rois.dim() == 3 and not gt_rois.requires_grad and not ex_rois.requires_grad:
        return ops.bbox_transform(ex_rois, gt_rois)            # one kernel instead of ~15
    ew, eh, ecx, ecy = _whc(ex_rois)
    gw, gh, gcx, gcy = _whc(gt_rois)
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), -1)


def bbox_transform(ex_rois, gt_rois):
    return bbox_transform_batch(ex_rois, gt_rois)


def bbox_transform_inv(boxes, deltas, batch_size=None):
    """(B,N,4) boxes + (B,N,4k) deltas -> (B,N,4k) predicted boxes (bbox_transform.py:77-103)."""
    w, h, cx, cy = (t.unsqueeze(2) for t in _whc(boxes))
    pcx = deltas[:, :, 0::4] * w + cx
    pcy = deltas[:, :, 1::4] * h + cy
    pw = torch.exp(deltas[:, :, 2::4]) * w
    ph = torch.exp(deltas[:, :, 3::4]) * h
    out = torch.empty_like(deltas)
    out[:, :, 0::4] = pcx - 0.5 * pw
    out[:, :, 1::4] = pcy - 0.5 * ph
    out[:, :, 2::4] = pcx + 0.5 * pw
    out[:, :, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape, batch_size=None):
    """In-place clamp to [0, w-1] x [0, h-1] per image (bbox_transform.py:125-133); im_shape (B,>=2) = [h, w, ..]."""
    xmax = (im_shape[:, 1] - 1).view(-1, 1, 1)
    ymax = (im_shape[:, 0] - 1).view(-1, 1, 1)
    zero = torch.zeros_like(xmax)
    boxes[:, :, 0::2] = torch.min(torch.max(boxes[:, :, 0::2], zero), xmax)
    boxes[:, :, 1::2] = torch.min(torch.max(boxes[:, :, 1::2], zero), ymax)
    return boxes


def bbox_overlaps_batch(anchors, gt_boxes):
    """(N,4) | (B,N,4) | (B,N,5) boxes vs (B,K,5) gt -> (B,N,K) IoU with the reference's zero-area
    masks (bbox_transform.py:168-257)."""
    return ops.bbox_overlaps(anchors, gt_boxes, want_matrix=True)[0]
