"""Seeded synthetic weights, frames, boxes and relation annotations.

No dataset or checkpoint is reachable (no network), so every parity fixture, the smoke
test and the benchmark are driven by the generators below.  Inputs are regenerated from a
seed with numpy's PCG64 ``default_rng`` (platform-stable), so golden files only need to
hold outputs.  Parameter names are the reference's state_dict keys
(``RCNN_base.6.3.conv2.weight``, ``vrd.fc6.fc.weight`` ...) so reference checkpoints
and these synthetic ones load through the same path.
"""
import math

import numpy as np
import torch

RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def _normal(rng, shape, std, device=None, gen=None):
    if rng is not None:
        return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(std)))
    return torch.randn(shape, device=device, generator=gen, dtype=torch.float32) * std


def _conv_w(rng, cout, cin, k, device, gen):
    # He-normal over fan-out as ResNet.__init__ does (resnet_instance_styleD_bilinear.py:238-241)
    return _normal(rng, (cout, cin, k, k), math.sqrt(2.0 / (k * k * cout)), device, gen)


def _bn(rng, c, device, gen, prefix, out, gamma=(0.5, 1.0)):
    if rng is not None:
        g = rng.uniform(gamma[0], gamma[1], c).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, c).astype(np.float32)
        m = rng.uniform(-0.1, 0.1, c).astype(np.float32)
        v = rng.uniform(0.5, 1.5, c).astype(np.float32)
        g, b, m, v = (torch.from_numpy(t) for t in (g, b, m, v))
    else:
        u = lambda lo, hi: torch.rand(c, device=device, generator=gen) * (hi - lo) + lo
        g, b, m, v = u(*gamma), u(-0.1, 0.1), u(-0.1, 0.1), u(0.5, 1.5)
    out[prefix + ".weight"], out[prefix + ".bias"] = g, b
    out[prefix + ".running_mean"], out[prefix + ".running_var"] = m, v


def _layer(rng, out, key, inplanes, planes, nblocks, stride, device, gen):
    for i in range(nblocks):
        k = "%s.%d" % (key, i)
        cin = inplanes if i == 0 else planes * 4
        out[k + ".conv1.weight"] = _conv_w(rng, planes, cin, 1, device, gen)
        _bn(rng, planes, device, gen, k + ".bn1", out)
        out[k + ".conv2.weight"] = _conv_w(rng, planes, planes, 3, device, gen)
        _bn(rng, planes, device, gen, k + ".bn2", out)
        out[k + ".conv3.weight"] = _conv_w(rng, planes * 4, planes, 1, device, gen)
        _bn(rng, planes * 4, device, gen, k + ".bn3", out, gamma=(0.2, 0.5))
        if i == 0 and (stride != 1 or inplanes != planes * 4):
            out[k + ".downsample.0.weight"] = _conv_w(rng, planes * 4, cin, 1, device, gen)
            _bn(rng, planes * 4, device, gen, k + ".downsample.1", out)
    return planes * 4


def backbone_params(seed=0, layers=101, top=True, device=None, numpy_rng=True):
    """conv1..layer3 as ``RCNN_base.N`` (+ layer4 as ``RCNN_top.0`` when ``top``)."""
    rng = np.random.default_rng(seed) if numpy_rng else None
    gen = None if numpy_rng else torch.Generator(device=device).manual_seed(seed)
    blocks = RESNET_BLOCKS[layers]
    p = {"RCNN_base.0.weight": _conv_w(rng, 64, 3, 7, device, gen)}
    _bn(rng, 64, device, gen, "RCNN_base.1", p)
    c = _layer(rng, p, "RCNN_base.4", 64, 64, blocks[0], 1, device, gen)
    c = _layer(rng, p, "RCNN_base.5", c, 128, blocks[1], 2, device, gen)
    c = _layer(rng, p, "RCNN_base.6", c, 256, blocks[2], 2, device, gen)
    if top:
        _layer(rng, p, "RCNN_top.0", c, 512, blocks[3], 2, device, gen)
    return _to(p, device)


def _to(p, device):
    if device is None:
        return p
    return {k: v.to(device) for k, v in p.items()}


def rpn_params(seed=10, din=1024, n_anchor=9, device=None, std=0.01):
    """rpn/rpn.py:27-36; normal(0, 0.01) like _init_weights (faster_rcnn_instance...:205-207)
    but with non-zero biases so the bias path is exercised."""
    rng = np.random.default_rng(seed)
    p = {
        "RCNN_rpn.RPN_Conv.weight": _normal(rng, (512, din, 3, 3), std),
        "RCNN_rpn.RPN_Conv.bias": _normal(rng, (512,), 0.01),
        "RCNN_rpn.RPN_cls_score.weight": _normal(rng, (2 * n_anchor, 512, 1, 1), std),
        "RCNN_rpn.RPN_cls_score.bias": _normal(rng, (2 * n_anchor,), 0.01),
        "RCNN_rpn.RPN_bbox_pred.weight": _normal(rng, (4 * n_anchor, 512, 1, 1), std),
        "RCNN_rpn.RPN_bbox_pred.bias": _normal(rng, (4 * n_anchor,), 0.01),
    }
    return _to(p, device)


def det_head_params(seed=11, n_classes=16, feat_d=2048, device=None):
    rng = np.random.default_rng(seed)
    p = {
        "RCNN_cls_score.weight": _normal(rng, (n_classes, feat_d), 0.01),
        "RCNN_cls_score.bias": _normal(rng, (n_classes,), 0.01),
        "RCNN_bbox_pred.weight": _normal(rng, (4 * n_classes, feat_d), 0.001),
        "RCNN_bbox_pred.bias": _normal(rng, (4 * n_classes,), 0.01),
    }
    return _to(p, device)


def netd_params(seed=12, dim=512, rank=5, device=None):
    """netD_pixel (1x1 convs, no bias) + netD_style (two 512->dim*rank FCs + 512->1)."""
    rng = np.random.default_rng(seed)
    p = {
        "netD_pixel.conv1.weight": _normal(rng, (512, 1024, 1, 1), 0.01),
        "netD_pixel.conv2.weight": _normal(rng, (128, 512, 1, 1), 0.01),
        "netD_pixel.conv3.weight": _normal(rng, (1, 128, 1, 1), 0.01),
        # kaiming_normal_(fan_out, relu): std = sqrt(2 / out_features)
        "netD_style.fc_1.weight": _normal(rng, (dim * rank, 512), math.sqrt(2.0 / (dim * rank))),
        "netD_style.fc_1.bias": _normal(rng, (dim * rank,), 0.02),
        "netD_style.fc_2.weight": _normal(rng, (dim * rank, 512), math.sqrt(2.0 / (dim * rank))),
        "netD_style.fc_2.bias": _normal(rng, (dim * rank,), 0.02),
        "netD_style.fc1.weight": _normal(rng, (1, dim), math.sqrt(2.0)),
        "netD_style.fc1.bias": _normal(rng, (1,), 0.02),
    }
    return _to(p, device)


VRD_SHAPES = (
    # key, out, in  (resnet_SGG_emb.py:83-127)
    ("vrd.fc6.fc", 4096, 1024 * 7 * 7),
    ("vrd.fc7.fc", 4096, 4096),
    ("vrd.so_vis_embeddings.fc", None, 4096),     # out = emb_dim
    ("vrd.fc8.fc", 256, 4096),
    ("vrd.fc_so.fc", 256, 600),
    ("vrd.fc_lov.fc", 256, 64),
    ("vrd.fc_fusion.fc", 256, 768),
    ("vrd.fc_rel.fc", None, 256),                 # out = emb_dim
    ("vrd.prd_sem_embeddings.0", 1024, 300),
    ("vrd.prd_sem_embeddings.2", None, 1024),     # out = emb_dim
)
VRD_CONVS = (("vrd.conv_lo.0.conv", 96, 2, 5), ("vrd.conv_lo.1.conv", 128, 96, 5),
             ("vrd.conv_lo.2.conv", 64, 128, 8))


def vrd_params(seed=13, emb_dim=300, device=None, numpy_rng=True, fc6_in=1024 * 7 * 7, use_obj_visual=True, spatial_type=2):
    """All ``vrd.*`` tensors: 226 480 996 parameters at emb_dim=300 (906 MB fp32).  ``use_obj_visual`` / ``spatial_type``: the
    variants of resnet_SGG_emb.py:94-123 (no ``fc_so``; ``fc_lov`` on 8 inputs / no spatial branch; ``fc_fusion`` narrower);
    the tensors every variant has are the same numbers as the default's (drawn in the same order, the others skipped)."""
    rng = np.random.default_rng(seed) if numpy_rng else None
    gen = None if numpy_rng else torch.Generator(device=device).manual_seed(seed)
    p = {}
    n_fusion = 256 * (1 + int(bool(use_obj_visual)) + int(spatial_type in (1, 2)))
    for key, cout, cin in VRD_SHAPES:
        cout = emb_dim if cout is None else cout
        cin = fc6_in if key == "vrd.fc6.fc" else cin
        full = (cout, cin)
        if key == "vrd.fc_lov.fc" and spatial_type == 1:
            cin = 8
        if key == "vrd.fc_fusion.fc":
            cin = n_fusion
        if (cout, cin) != full or (key == "vrd.fc_so.fc" and not use_obj_visual) or (key == "vrd.fc_lov.fc" and spatial_type not in (1, 2)):
            # keep the default's random stream for every other tensor: draw the default-shaped tensor, then a narrower one
            _normal(rng, full, 1.0 / math.sqrt(full[1]), device, gen)
            _normal(rng, (full[0],), 1.0 / math.sqrt(full[1]), device, gen)
            if (key == "vrd.fc_so.fc" and not use_obj_visual) or (key == "vrd.fc_lov.fc" and spatial_type not in (1, 2)):
                continue
            r2 = np.random.default_rng(seed + 1000 + cin) if numpy_rng else None
            bound = 1.0 / math.sqrt(cin)
            p[key + ".weight"] = _normal(r2, (cout, cin), bound, device, gen)
            p[key + ".bias"] = _normal(r2, (cout,), bound, device, gen)
            continue
        bound = 1.0 / math.sqrt(cin)                 # nn.Linear default scale
        p[key + ".weight"] = _normal(rng, (cout, cin), bound, device, gen)
        p[key + ".bias"] = _normal(rng, (cout,), bound, device, gen)
    for key, cout, cin, k in VRD_CONVS:
        bound = 1.0 / math.sqrt(cin * k * k)
        w, b = _normal(rng, (cout, cin, k, k), bound, device, gen), _normal(rng, (cout,), bound, device, gen)
        if spatial_type == 2:
            p[key + ".weight"], p[key + ".bias"] = w, b
    return _to(p, device)


# ------------------------------------------------------------------------ inputs
def frames(seed, batch, h=600, w=1000):
    """Mean-subtracted BGR-scale frames (B,3,h,w) fp32 and im_info (B,3)=[h,w,1]."""
    rng = np.random.default_rng(seed)
    im = (rng.standard_normal((batch, 3, h, w), dtype=np.float32) * np.float32(50.0))
    info = np.tile(np.array([[h, w, 1.0]], np.float32), (batch, 1))
    return im, info


def boxes(seed, n, im_h=600, im_w=1000, min_side=32, max_side=400):
    """n boxes: x1,y1 uniform, w,h in [min_side,max_side] px, clipped (fp32, (n,4))."""
    rng = np.random.default_rng(seed)
    x1 = rng.uniform(0, im_w - min_side - 1, n)
    y1 = rng.uniform(0, im_h - min_side - 1, n)
    bw = rng.uniform(min_side, max_side, n)
    bh = rng.uniform(min_side, max_side, n)
    x2 = np.minimum(x1 + bw, im_w - 1)
    y2 = np.minimum(y1 + bh, im_h - 1)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)


def gt_boxes(seed, batch, n_gt, n_classes=16, max_gt=30, im_h=600, im_w=1000):
    """(B,max_gt,5) zero-padded [x1,y1,x2,y2,cls] with integer pixel corners, num_boxes (B,)."""
    out = np.zeros((batch, max_gt, 5), np.float32)
    rng = np.random.default_rng(seed + 7919)
    for b in range(batch):
        bx = np.floor(boxes(seed * 131 + b, n_gt, im_h, im_w, 48, 360))
        out[b, :n_gt, :4] = bx
        out[b, :n_gt, 4] = rng.integers(1, n_classes, n_gt)
    return out, np.full((batch,), n_gt, np.int64)


def relation_annotation(seed, n_boxes=32, n_pairs=32, n_rel=62, n_classes=16, im_h=600, im_w=1000):
    """One frame's VidVRD-style annotation dict like ``vrd.source_gt_rels[im_path]``
    (faster_rcnn_SGG_emb.py:169-172): boxes (python lists, unscaled pixels), box_classes,
    rels = [s, o, predicate] with 1-3 predicates on each of n_pairs distinct ordered pairs."""
    rng = np.random.default_rng(seed)
    bx = np.floor(boxes(seed + 1, n_boxes, im_h, im_w)).astype(np.float64)
    pairs = set()
    while len(pairs) < n_pairs:
        s, o = (int(v) for v in rng.integers(0, n_boxes, 2))
        if s != o:
            pairs.add((s, o))
    rels = []
    for s, o in sorted(pairs, key=lambda _: rng.random()):
        for r in rng.choice(n_rel, size=int(rng.integers(1, 4)), replace=False):
            rels.append([s, o, int(r)])
    order = rng.permutation(len(rels))
    return {"boxes": bx.tolist(), "box_classes": rng.integers(1, n_classes, n_boxes).tolist(),
            "rels": [rels[i] for i in order]}


def word_vectors(seed, n, dim=300):
    """Stand-in for the GloVe rows (``all_obj_vecs`` / ``all_prd_vecs``): (n,dim) fp32."""
    return np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)


def tie_free_dets(seed, n, im_h=600, im_w=1000, clustered=False):
    """(n,5) [x1,y1,x2,y2,score] sorted by strictly decreasing score (tie-free)."""
    rng = np.random.default_rng(seed)
    if clustered:
        k = max(1, n // 40)
        centres = boxes(seed + 3, k, im_h, im_w, 48, 300)
        b = centres[rng.integers(0, k, n)] + rng.normal(0, 6.0, (n, 4)).astype(np.float32)
        b = np.stack([np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3]),
                      np.maximum(b[:, 0], b[:, 2]), np.maximum(b[:, 1], b[:, 3])], 1)
        b[:, 0::2] = np.clip(b[:, 0::2], 0, im_w - 1)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, im_h - 1)
    else:
        b = boxes(seed + 3, n, im_h, im_w, 16, 400)
    s = np.sort(rng.permutation(4 * n + 16)[:n].astype(np.float32))[::-1] / np.float32(4 * n + 16)
    return np.concatenate([b.astype(np.float32), s[:, None].astype(np.float32)], 1)


def roi_cases(seed, B, H, W, scale=16.0, n=24):
    """(n+6,5) ROIs [b,x1,y1,x2,y2] for a (B,C,H,W) map: n seeded boxes plus a full-image, a sub-pixel, a malformed
    (x2<x1, y2<y1), a partly outside (negative), a beyond-the-far-edge and a single-point box -- the classes
    tests/golden/roi_align_fwd.npz (tools/gen_golden.py gen_roi_align) and the ROI kernel tests cover."""
    rng = np.random.default_rng(seed)
    bx = boxes(int(rng.integers(1 << 30)), n, H * scale, W * scale, 16, min(H, W) * scale * 0.9)
    r = np.zeros((n + 6, 5), np.float32)
    r[:n, 1:] = bx
    r[:n, 0] = rng.integers(0, B, n)
    r[n + 0] = [0, 0, 0, W * scale - 1, H * scale - 1]            # full image
    r[n + 1] = [B - 1, 40.5, 33.25, 41.0, 33.5]                   # sub-pixel
    r[n + 2] = [0, 120, 90, 60, 30]                               # malformed: x2<x1, y2<y1
    r[n + 3] = [B - 1, -50, -40, 80, 70]                          # partly outside (negative)
    r[n + 4] = [0, W * scale - 30, H * scale - 30, W * scale + 90, H * scale + 60]   # beyond the far edge
    r[n + 5] = [B - 1, 10, 10, 10, 10]                            # single point
    return r


ROI_ALIGN_GOLDEN_CASES = ((4, 9, 11, 2), (64, 19, 32, 2), (1024, 38, 63, 1))          # (C, H, W, B)


def roi_align_golden_inputs(C, H, W, B):
    """The seeded map (B,C,H,W) and rois of one tests/golden/roi_align_fwd.npz case."""
    feat = np.random.default_rng(9000 + C + H).standard_normal((B, C, H, W), dtype=np.float32)
    return feat, roi_cases(9100 + C, B, H, W)


def instance_styled_step_params(layers=101, n_cls=16):
    """Seeded weights of the tests/golden/instance_styled_step.npz model (reference state_dict keys)."""
    p = {}
    p.update(backbone_params(0, layers, top=True))
    p.update(rpn_params(10, std=0.01))      # 0.02 saturates the scores: hundreds of exact ties among the top proposals
    p.update(det_head_params(11, n_cls))
    p.update(netd_params(12))
    return p


def instance_styled_step_inputs(B, H, W, n_cls=16):
    """Source frames + gt, target frames of one instance_styleD step of the golden: (im, info, gt, nb, im_t, info_t)."""
    im, info = frames(5, B, H, W)
    gt, nb = gt_boxes(6, B, 6, n_cls, im_h=H, im_w=W)
    im_t, info_t = frames(8, B, H, W)
    return im, info, gt, nb, im_t, info_t


def context_inputs(B=2, H=320, W=480):
    """Inputs of the ic/gc fixture (tools/gen_golden.py gen_context and tests/test_gpu_models.py): frames, gt, fixed proposals."""
    im, info = frames(5, B, H, W)
    gt, nb = gt_boxes(6, B, 6, 16, im_h=H, im_w=W)
    rois = np.zeros((B, 600, 5), np.float32)
    for b in range(B):
        rois[b, :, 0] = b
        rois[b, :, 1:] = boxes(430 + b, 600, im_h=H, im_w=W, min_side=16, max_side=220)
        jit = np.random.default_rng(440 + b).normal(0, 6, (48, 4)).astype(np.float32)
        rois[b, :48, 1:] = np.clip(gt[b, np.arange(48) % 6, :4] + jit, 0, [W - 1, H - 1, W - 1, H - 1])
    return im, info, gt, nb, rois
