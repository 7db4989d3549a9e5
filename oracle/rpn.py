"""numpy restatement of the RPN host logic (TEST INFRASTRUCTURE -- see oracle/__init__.py).

All box arithmetic is fp32 with one rounding per op, like the reference's torch-CPU
tensors.  Citations are into /root/reference/lib/model.
"""
import numpy as np

from . import cops

F32 = np.float32


# --------------------------------------------------------------------------- anchors
def base_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """rpn/generate_anchors.py:45-56 -> (len(ratios)*len(scales), 4) float64, ratio-major."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    ctr = 0.5 * (base_size - 1)                       # window (0,0,15,15) -> centre 7.5
    area = float(base_size * base_size)
    ws = np.round(np.sqrt(area / ratios))             # :85-88  (np.round = half-to-even)
    hs = np.round(ws * ratios)
    rows = []
    for w, h in zip(ws, hs):
        for s in scales:                              # :94-103
            hw, hh = 0.5 * (w * s - 1), 0.5 * (h * s - 1)
            rows.append([ctr - hw, ctr - hh, ctr + hw, ctr + hh])
    return np.array(rows, dtype=np.float64)


def anchor_grid(feat_h, feat_w, stride=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """rpn/proposal_layer.py:80-95: (y, x, a) row-major grid -> (H*W*A, 4) fp32."""
    base = base_anchors(stride, ratios, scales).astype(F32)
    sx, sy = np.meshgrid(np.arange(feat_w) * stride, np.arange(feat_h) * stride)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], 1).astype(F32)
    return (base[None, :, :] + shifts[:, None, :]).reshape(-1, 4)


# --------------------------------------------------------------------------- box math
def decode_clip(anchors, deltas, im_h, im_w):
    """rpn/bbox_transform.py:77-103 + :125-133 (one image)."""
    return cops.decode_clip(anchors, deltas, float(im_h), float(im_w))


def iou_matrix(boxes, gt):
    """rpn/bbox_transform.py:168-213 for one image: boxes (N,4), gt (K,>=4) -> (N,K) fp32.

    Zero-area gt (w==1 and h==1, i.e. zero padding) -> 0; zero-area box -> -1.
    """
    boxes = np.asarray(boxes, F32)
    gt = np.asarray(gt, F32)[:, :4]
    gw = gt[:, 2] - gt[:, 0] + F32(1)
    gh = gt[:, 3] - gt[:, 1] + F32(1)
    bw = boxes[:, 2] - boxes[:, 0] + F32(1)
    bh = boxes[:, 3] - boxes[:, 1] + F32(1)
    garea = (gw * gh)[None, :]
    barea = (bw * bh)[:, None]
    iw = np.minimum(boxes[:, None, 2], gt[None, :, 2]) - np.maximum(boxes[:, None, 0], gt[None, :, 0]) + F32(1)
    ih = np.minimum(boxes[:, None, 3], gt[None, :, 3]) - np.maximum(boxes[:, None, 1], gt[None, :, 1]) + F32(1)
    iw = np.where(iw < 0, F32(0), iw)
    ih = np.where(ih < 0, F32(0), ih)
    inter = iw * ih
    ov = inter / (barea + garea - inter)
    ov = np.where(((gw == 1) & (gh == 1))[None, :], F32(0), ov)
    ov = np.where(((bw == 1) & (bh == 1))[:, None], F32(-1), ov)
    return ov.astype(F32)


def box_targets(ex, gt):
    """rpn/bbox_transform.py:36-75: regression targets of ex (N,4) towards gt (N,4)."""
    ex = np.asarray(ex, F32)
    gt = np.asarray(gt, F32)
    ew = ex[:, 2] - ex[:, 0] + F32(1)
    eh = ex[:, 3] - ex[:, 1] + F32(1)
    ecx = ex[:, 0] + F32(0.5) * ew
    ecy = ex[:, 1] + F32(0.5) * eh
    gw = gt[:, 2] - gt[:, 0] + F32(1)
    gh = gt[:, 3] - gt[:, 1] + F32(1)
    gcx = gt[:, 0] + F32(0.5) * gw
    gcy = gt[:, 1] + F32(0.5) * gh
    return np.stack([(gcx - ecx) / ew, (gcy - ecy) / eh,
                     np.log(gw / ew), np.log(gh / eh)], 1).astype(F32)


# --------------------------------------------------------------------------- proposal layer
def proposal_layer(fg_scores, deltas, im_info, pre_nms_top_n, post_nms_top_n, nms_thresh,
                   stride=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """rpn/proposal_layer.py:49-163.

    fg_scores (B,A,H,W) = rpn_cls_prob[:, A:], deltas (B,4A,H,W), im_info (B,3).
    Returns rois (B, post_nms_top_n, 5), zero padded, col 0 = image index, plus the
    per-image kept anchor indices (for index-exact parity checks).
    Order rule on ties: descending score, ascending anchor index (stable sort); the
    reference's order on exact ties is unspecified.
    """
    B, A, H, W = fg_scores.shape
    anchors = anchor_grid(H, W, stride, ratios, scales)
    out = np.zeros((B, post_nms_top_n, 5), dtype=F32)
    kept = []
    for b in range(B):
        sc = np.ascontiguousarray(fg_scores[b].transpose(1, 2, 0)).reshape(-1)          # :103-104
        dl = np.ascontiguousarray(deltas[b].transpose(1, 2, 0)).reshape(-1, 4)          # :99-100
        props = decode_clip(anchors, dl, im_info[b, 0], im_info[b, 1])
        order = np.argsort(-sc, kind="stable")                                          # :127
        if 0 < pre_nms_top_n < B * sc.size:                                             # :140
            order = order[:pre_nms_top_n]
        dets = np.concatenate([props[order], sc[order, None]], 1)
        keep = cops.nms_sorted(dets, nms_thresh)                                        # :150
        if post_nms_top_n > 0:
            keep = keep[:post_nms_top_n]
        out[b, :, 0] = b
        out[b, :keep.size, 1:] = dets[keep, :4]
        kept.append(order[keep])
    return out, kept


# --------------------------------------------------------------------------- anchor targets
def anchor_target_layer(feat_h, feat_w, gt_boxes, im_info, rng, *, rpn_batch=256, fg_frac=0.5,
                        pos_ov=0.7, neg_ov=0.3, stride=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """rpn/anchor_target_layer.py:48-193.  gt_boxes (B,G,5); `rng` is an object with
    numpy's legacy ``permutation`` (np.random or a RandomState) -- the call order is
    part of the contract (:131, :143).  Returns labels (B,1,A*H,W), targets,
    inside-w, outside-w (B,4A,H,W)."""
    gt_boxes = np.asarray(gt_boxes, F32)
    B = gt_boxes.shape[0]
    A = len(ratios) * len(scales)
    allanc = anchor_grid(feat_h, feat_w, stride, ratios, scales)
    total = allanc.shape[0]
    imw, imh = int(im_info[0][1]), int(im_info[0][0])                                   # :85-86 (image 0)
    inside = np.nonzero((allanc[:, 0] >= 0) & (allanc[:, 1] >= 0) &
                        (allanc[:, 2] < imw) & (allanc[:, 3] < imh))[0]
    anc = allanc[inside]
    n = inside.size
    labels = np.full((B, n), -1, dtype=F32)
    ov = np.stack([iou_matrix(anc, gt_boxes[b]) for b in range(B)])                     # (B,n,G)
    max_ov = ov.max(2)
    argmax = ov.argmax(2)
    gt_max = ov.max(1)                                                                  # (B,G)
    labels[max_ov < neg_ov] = 0                                                         # :105-106
    gt_max = np.where(gt_max == 0, F32(1e-5), gt_max)                                   # :108
    hit = (ov == gt_max[:, None, :]).sum(2)
    labels[hit > 0] = 1                                                                 # :111-112
    labels[max_ov >= pos_ov] = 1                                                        # :115
    num_fg = int(fg_frac * rpn_batch)
    sum_fg = (labels == 1).sum(1)
    sum_bg = (labels == 0).sum(1)
    for b in range(B):                                                                  # :123-147
        if sum_fg[b] > num_fg:
            fg = np.nonzero(labels[b] == 1)[0]
            perm = rng.permutation(fg.size)
            labels[b, fg[perm[:fg.size - num_fg]]] = -1
        num_bg = rpn_batch - int((labels[b] == 1).sum())
        if sum_bg[b] > num_bg:
            bg = np.nonzero(labels[b] == 0)[0]
            perm = rng.permutation(bg.size)
            labels[b, bg[perm[:bg.size - num_bg]]] = -1
    tg = np.stack([box_targets(anc, gt_boxes[b][argmax[b], :4]) for b in range(B)])     # :151-152
    inw = np.where(labels == 1, F32(1.0), F32(0.0))                                     # :155
    num_examples = int((labels[B - 1] >= 0).sum())                                      # :157-160 (last image)
    wgt = F32(1.0 / num_examples)
    outw = np.where(labels >= 0, wgt, F32(0.0)).astype(F32)

    def unmap(x, fill):
        full = np.full((B, total) + x.shape[2:], fill, dtype=F32)
        full[:, inside] = x
        return full

    L = unmap(labels, -1).reshape(B, feat_h, feat_w, A).transpose(0, 3, 1, 2).reshape(B, 1, A * feat_h, feat_w)
    T = unmap(tg, 0).reshape(B, feat_h, feat_w, 4 * A).transpose(0, 3, 1, 2)
    IW = np.repeat(unmap(inw, 0)[:, :, None], 4, 2).reshape(B, feat_h, feat_w, 4 * A).transpose(0, 3, 1, 2)
    OW = np.repeat(unmap(outw, 0)[:, :, None], 4, 2).reshape(B, feat_h, feat_w, 4 * A).transpose(0, 3, 1, 2)
    return (np.ascontiguousarray(L), np.ascontiguousarray(T),
            np.ascontiguousarray(IW), np.ascontiguousarray(OW))


# --------------------------------------------------------------------------- proposal targets
def proposal_target_layer(all_rois, gt_boxes, rng, *, batch_size=128, fg_fraction=0.25,
                          fg_thresh=0.5, bg_hi=0.5, bg_lo=0.0,
                          means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2)):
    """rpn/proposal_target_layer_cascade.py:33-56,116-212.  all_rois (B,P,5),
    gt_boxes (B,G,5).  `rng` provides numpy's legacy ``permutation`` and ``rand``
    (call order :158, :167, :174, :182).  Returns rois (B,R,5), labels (B,R),
    targets, inside-w, outside-w (B,R,4)."""
    all_rois = np.asarray(all_rois, F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    B, G = gt_boxes.shape[:2]
    app = np.zeros_like(gt_boxes)
    app[:, :, 1:5] = gt_boxes[:, :, :4]
    rois_all = np.concatenate([all_rois, app], 1)                                       # :41-45
    R = int(batch_size)
    fg_per = int(np.round(fg_fraction * R)) or 1
    labels_b = np.zeros((B, R), F32)
    rois_b = np.zeros((B, R, 5), F32)
    gt_b = np.zeros((B, R, 5), F32)
    for b in range(B):
        ov = iou_matrix(rois_all[b, :, 1:5], gt_boxes[b])
        max_ov, assign = ov.max(1), ov.argmax(1)
        lab = gt_boxes[b, assign, 4]
        fg = np.nonzero(max_ov >= fg_thresh)[0]
        bg = np.nonzero((max_ov < bg_hi) & (max_ov >= bg_lo))[0]
        if fg.size > 0 and bg.size > 0:                                                 # :151-168
            nfg = min(fg_per, fg.size)
            fg = fg[rng.permutation(fg.size)[:nfg]]
            nbg = R - nfg
            bg = bg[np.floor(rng.rand(nbg) * bg.size).astype(np.int64)]
        elif fg.size > 0:                                                               # :170-177
            fg = fg[np.floor(rng.rand(R) * fg.size).astype(np.int64)]
            nfg, bg = R, bg[:0]
        elif bg.size > 0:                                                               # :178-186
            bg = bg[np.floor(rng.rand(R) * bg.size).astype(np.int64)]
            nfg, fg = 0, fg[:0]
        else:
            raise ValueError("no fg and no bg rois")
        keep = np.concatenate([fg, bg])
        labels_b[b] = lab[keep]
        if nfg < R:
            labels_b[b, nfg:] = 0
        rois_b[b] = rois_all[b, keep]
        rois_b[b, :, 0] = b
        gt_b[b] = gt_boxes[b, assign[keep]]
    tg = np.stack([box_targets(rois_b[b, :, 1:5], gt_b[b, :, :4]) for b in range(B)])
    tg = ((tg - np.asarray(means, F32)) / np.asarray(stds, F32)).astype(F32)            # :107-110
    fgmask = (labels_b > 0)[:, :, None]
    targets = np.where(fgmask, tg, F32(0)).astype(F32)                                  # :72-91
    inw = np.where(fgmask, F32(1.0), F32(0.0)) * np.ones(4, F32)
    outw = (inw > 0).astype(F32)
    return rois_b, labels_b, targets, inw.astype(F32), outw


# --------------------------------------------------------------------------- losses
def smooth_l1(pred, target, in_w, out_w, sigma=1.0, sum_dims=(1,)):
    """utils/net_utils.py:122-136 (numpy, fp32)."""
    s2 = F32(sigma * sigma)
    d = (in_w * (pred - target)).astype(F32)
    ad = np.abs(d)
    small = (ad < F32(1.0) / s2).astype(F32)
    loss = d * d * (s2 / F32(2)) * small + (ad - F32(0.5) / s2) * (F32(1) - small)
    loss = out_w * loss
    for ax in sorted(sum_dims, reverse=True):
        loss = loss.sum(ax)
    return F32(loss.mean())


def detection_postprocess(rois, cls_prob, bbox_pred, im_h, im_w, im_scale, class_agnostic=False, stds=None, means=None,
                          score_thresh=0.0, nms_thresh=0.3, max_per_image=100):
    """test_net_instance_styleD_bilinear.py:151-221 for one image.  Returns the reference's ``all_boxes[j][i]`` as a
    list over classes (entry 0 = background, empty).  Sort rule on tied scores: descending score, ascending roi
    index (torch.sort's order on exact ties is unspecified)."""
    rois = np.asarray(rois, np.float32).reshape(-1, 5)
    R = rois.shape[0]
    scores = np.asarray(cls_prob, np.float32).reshape(R, -1)
    C = scores.shape[1]
    deltas = np.asarray(bbox_pred, np.float32).reshape(-1, 4)            # .view(-1, 4)  :158,:161
    if stds is not None:
        deltas = (deltas * np.asarray(stds, np.float32)).astype(np.float32) + np.asarray(means, np.float32)
        deltas = deltas.astype(np.float32)
    deltas = deltas.reshape(R, -1)
    boxes = rois[:, 1:5]
    out = [np.zeros((0, 5), np.float32)]
    for j in range(1, C):
        d = deltas if class_agnostic else deltas[:, 4 * j:4 * j + 4]
        pred = decode_clip(boxes, d, im_h, im_w)                         # bbox_transform_inv + clip_boxes :165-166
        pred = (pred / np.float32(im_scale)).astype(np.float32)          # :171
        inds = np.nonzero(scores[:, j] > np.float32(score_thresh))[0]    # :182
        if inds.size == 0:
            out.append(np.zeros((0, 5), np.float32))
            continue
        cs = scores[inds, j]
        order = np.argsort(-cs, kind="stable")
        dets = np.concatenate([pred[inds], cs[:, None]], 1).astype(np.float32)[order]
        keep = cops.nms_sorted(dets, nms_thresh)                         # nms(cls_dets, cfg.TEST.NMS) :195
        out.append(dets[keep])
    if max_per_image > 0:                                                # :214-221
        image_scores = np.hstack([out[j][:, -1] for j in range(1, C)])
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, C):
                out[j] = out[j][out[j][:, -1] >= image_thresh]
    return out
