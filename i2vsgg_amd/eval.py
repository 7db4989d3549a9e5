"""Per-frame evaluation helpers: what the reference's test scripts do between the model forward and the pickle.

``detect_frame``  test_net_instance_styleD_bilinear.py:140-221 -- eval forward, then the per-class detection
                  post-processing as ONE device pass (``ops.detection_postprocess``; the reference makes
                  n_classes - 1 host NMS calls per frame) and a single D2H copy of the surviving boxes.
``detection_output``  lib/utils.py:584-628 -- top-100 relation triplets of a frame from the ``vrd_data`` dict the
                  eval branch of ``forward_relation`` returns; scaling by the box confidences, the ranking over the
                  (pair, predicate) grid and the top-k cut run on the device (``ops.relation_topk``).
``relation_frame``  backbone forward + ``forward_relation_eval`` + ``detection_output`` for one frame.
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


def detection_output(vrd_data, k=100):
    """lib/utils.py:584-628.  Returns (rlp_labels_im (100,3), tuple_confs_im (k',), sub_bboxes_im (100,4),
    obj_bboxes_im (100,4), rel_idex (k',)) or five Nones when the frame has fewer than two boxes."""
    if len(vrd_data["bboxes"]) <= 1:
        return None, None, None, None, None
    ixs, ixo = np.asarray(vrd_data["ixs"]), np.asarray(vrd_data["ixo"])
    boxes, classes = np.asarray(vrd_data["bboxes"], np.float64), np.asarray(vrd_data["classes"])
    rel = vrd_data["rel_score"]
    dev = rel.device
    conf = torch.as_tensor(np.asarray(vrd_data["scores"], np.float32), device=dev)
    pair, pred, tconf = ops.relation_topk(rel, conf, torch.as_tensor(ixs, device=dev), torch.as_tensor(ixo, device=dev), k)
    pair, pred, tconf = pair.cpu().numpy(), pred.cpu().numpy(), tconf.cpu().numpy()      # <= 100 rows
    n = pair.shape[0]
    rlp = np.zeros((k, 3), np.float64)
    sub = np.zeros((k, 4), np.float64)
    obj = np.zeros((k, 4), np.float64)
    sub[:n], obj[:n] = boxes[ixs[pair]], boxes[ixo[pair]]
    rlp[:n] = np.stack([classes[ixs[pair]], pred, classes[ixo[pair]]], 1)
    return rlp, tconf, sub, obj, pair.astype(np.int64)


@torch.no_grad()
def relation_frame(net, im_data, im_info, im_path, k=100):
    """test_net_SGG_emb.py per-frame work: backbone, eval relation branch, top-k triplets."""
    fmap = net.RCNN_base(im_data)
    vrd_data = net.forward_relation_eval(fmap, im_info, im_path)
    return vrd_data, detection_output(vrd_data, k)
