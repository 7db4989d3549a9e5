"""Per-frame evaluation helpers: what the reference's test scripts do between the model forward and the pickle.

``detect_frame``  test_net_instance_styleD_bilinear.py:140-221 -- eval forward, then the per-class detection
                  post-processing as ONE device pass (``ops.detection_postprocess``; the reference makes
                  n_classes - 1 host NMS calls per frame) and a single D2H copy of the surviving boxes.
"""
import numpy as np
import torch

from . import ops
from .model.utils.config import cfg


@torch.no_grad()
def detect_frame(net, im_data, im_info, gt_boxes, num_boxes, thresh=0.0, max_per_image=100, class_agnostic=False):
    """Returns ``all_boxes_i``: list over classes of (n_j, 5) float32 arrays [x1,y1,x2,y2,score] in original-image
    coordinates (entry 0, background, is empty) -- the reference's ``all_boxes[j][i]`` for this frame."""
    assert im_data.shape[0] == 1, "the reference evaluates one frame at a time (test_net_...:95 batch_size 1)"
    out = net(im_data, im_info, gt_boxes, num_boxes)
    rois, cls_prob, bbox_pred = out[0], out[1], out[2]
    info = im_info.reshape(-1).tolist()
    stds = means = None
    if cfg.TEST.BBOX_REG and cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
        stds, means = cfg.TRAIN.BBOX_NORMALIZE_STDS, cfg.TRAIN.BBOX_NORMALIZE_MEANS
    R, C = cls_prob.shape[-2], cls_prob.shape[-1]
    if not cfg.TEST.BBOX_REG:                      # :167-169 "simply repeat the boxes": zero deltas, no de-normalisation
        bbox_pred = torch.zeros((R, 4), device=rois.device)
        class_agnostic, stds, means = True, None, None
    dets, counts = ops.detection_postprocess(rois, cls_prob, bbox_pred, info[0], info[1], info[2], class_agnostic, stds, means,
                                             thresh, cfg.TEST.NMS, max_per_image)
    counts = counts.cpu().numpy()                  # the one synchronisation of the frame
    dets = dets.cpu().numpy()
    return [np.ascontiguousarray(dets[j, :counts[j]]) for j in range(C)]
