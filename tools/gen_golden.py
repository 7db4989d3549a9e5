#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python in this container.

Build-container only: needs /root/reference (read-only) and never ships to the GPU box;
only the .npz outputs do.  Inputs are regenerated from seeds by i2vsgg_amd.synthetic, so
the fixtures hold reference OUTPUTS (plus the few inputs that are cheaper to store than to
regenerate).

Two tiers, recorded per file in the ``tier`` field:
  "direct"       the reference module imports as shipped (sys.path only):
                 rpn/generate_anchors.py, rpn/bbox_transform.py, nms/nms_cpu.py
  "extracted"    roi_align/src/roi_align.c:80-136 ``ROIAlignForwardCpu``: the file as a whole needs <TH/TH.h> (its
                 THFloatTensor wrappers), the function itself only <math.h>; oracle/build_ref.py pipes the function's
                 own text to gcc unmodified (no stand-in header, nothing of it written to disk but the .so)
                 lib/utils.py:584-628 ``detection_output``: the module cannot be imported (a module-level json.load of an
                 absolute path), so the function's own lines are compiled from the file and run unmodified, with the
                 ``np.float`` alias numpy 1.24 removed restored for the call
  "placeholders" the module imports after registering in-process placeholders for
                 three third-party packages absent from this image and for the
                 reference's un-buildable compiled extensions:
                   easydict.EasyDict  -> attribute-access dict (what the package is)
                   torchvision, cv2   -> empty modules (imported, unused on this path)
                   model.<ext>._ext   -> empty modules (torch.utils.ffi builds, gone in torch>=1.0)
                   model._C           -> roi_pool_forward backed by oracle.cops (vrd only;
                                         the real source is absent from the reference tree)
                 The reference's own code runs unmodified.
"""
import argparse
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "lib"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from i2vsgg_amd import synthetic as syn  # noqa: E402
from oracle import cops  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def save(name, tier, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, tier=np.array(tier), **arrays)
    print("  wrote %-38s %8.1f KB  [%s]" % (name + ".npz", os.path.getsize(path) / 1024.0, tier))


# ----------------------------------------------------------------------------- tier "direct"
def gen_direct():
    from model.rpn.generate_anchors import generate_anchors
    from model.rpn import bbox_transform as bt
    from model.nms.nms_cpu import nms_cpu

    base = generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))
    # the shift-grid recipe of proposal_layer.py:81-95, executed with the reference's base anchors
    H, W, stride = 38, 63, 16
    sx, sy = np.meshgrid(np.arange(0, W) * stride, np.arange(0, H) * stride)
    shifts = torch.from_numpy(np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()).contiguous().float()
    anc = (torch.from_numpy(base).float().view(1, 9, 4) + shifts.view(-1, 1, 4)).view(-1, 4)
    save("anchors", "direct", base=base, grid_38x63=anc.numpy())

    # decode + clip on seeded deltas (B=2 so that per-image clip limits differ)
    rng = np.random.default_rng(101)
    deltas = (rng.standard_normal((2, anc.shape[0], 4)) * np.array([0.3, 0.3, 0.5, 0.5])).astype(np.float32)
    im_info = np.array([[600, 1000, 1.0], [576, 992, 1.2]], np.float32)
    boxes = anc.view(1, -1, 4).expand(2, -1, 4)
    prop = bt.bbox_transform_inv(boxes, torch.from_numpy(deltas), 2)
    prop = bt.clip_boxes(prop, torch.from_numpy(im_info), 2)
    save("decode_clip", "direct", im_info=im_info, proposals=prop.numpy())

    # IoU and regression targets
    gt, _ = syn.gt_boxes(5, 2, 8)
    rois = np.stack([syn.boxes(50 + b, 300) for b in range(2)])
    rois[0, 5] = [10, 10, 10, 10]                      # zero-area roi -> -1 row
    ov3 = bt.bbox_overlaps_batch(torch.from_numpy(rois), torch.from_numpy(gt))
    ov2 = bt.bbox_overlaps_batch(anc[::7].contiguous(), torch.from_numpy(gt))
    ex = torch.from_numpy(rois)
    gts = torch.from_numpy(np.stack([syn.boxes(70 + b, 300) for b in range(2)]))
    tg = bt.bbox_transform_batch(ex, gts)
    save("box_math", "direct", rois=rois, gt=gt, overlaps_rois=ov3.numpy(), overlaps_anchors=ov2.numpy(),
         tgt_gt=gts.numpy(), targets=tg.numpy())

    # NMS keep lists on tie-free, pre-sorted dets
    out = {}
    for n in (1, 2, 63, 64, 65, 300, 1000, 6000, 12000):
        for clustered in (False, True):
            dets = syn.tie_free_dets(1000 + n, n, clustered=clustered)
            for th in (0.7, 0.3):
                keep = nms_cpu(torch.from_numpy(dets), th).numpy().astype(np.int32)
                out["n%d_%s_t%02d" % (n, "c" if clustered else "u", int(th * 10))] = keep
    save("nms_keep", "direct", **out)


# ----------------------------------------------------------------------------- RoIAlign: the reference's own C (tier "extracted")
def gen_roi_align():
    """``ROIAlignForwardCpu`` (roi_align/src/roi_align.c:80-136) -- the function's own text compiled
    where it lies by oracle/build_ref.py (self-contained: <math.h> only; the THFloatTensor wrappers at :16-78 are not
    built) -- on 8x8 grids, and ``RoIAlignAvg``'s ``avg_pool2d(x, kernel_size=2, stride=1)`` of it
    (roi_align/modules/roi_align.py:27-29) -> 7x7, for C in {4, 64, 1024} over ROIs of every class (seeded, full-image,
    sub-pixel, malformed, partly outside, beyond the far edge, single point).  Full tensors up to C = 64; the C = 1024 case
    (7.9 MB) as sha256 of the bytes + sums + a strided sample."""
    import hashlib
    from oracle import build_ref
    build_ref.build()
    out = {}
    for C, H, W, B in syn.ROI_ALIGN_GOLDEN_CASES:
        feat, rois = syn.roi_align_golden_inputs(C, H, W, B)
        a8 = build_ref.roi_align_fwd(feat, rois, 8, 8, 1.0 / 16.0)
        a7 = torch.nn.functional.avg_pool2d(torch.from_numpy(a8), kernel_size=2, stride=1).numpy()
        p7 = build_ref.roi_align_fwd(feat, rois, 7, 7, 1.0 / 16.0)           # RoIAlign(7,7): the un-averaged module
        tag = "c%d" % C
        out[tag + "_rois"] = rois
        for key, v in (("a8", a8), ("avg7", a7), ("p7", p7)):
            out["%s_%s_shape" % (tag, key)] = np.array(v.shape)
            out["%s_%s_sha256" % (tag, key)] = np.array(hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest())
            out["%s_%s_sum" % (tag, key)] = np.array(v.astype(np.float64).sum())
            out["%s_%s_abs" % (tag, key)] = np.array(np.abs(v.astype(np.float64)).sum())
            if C <= 64:
                out["%s_%s" % (tag, key)] = v
            else:
                out["%s_%s_sample" % (tag, key)] = v.reshape(-1)[::211].copy()
        same = np.array_equal(a8, cops.roi_align_fwd(feat, rois, 8, 8, 1.0 / 16.0))
        print("   C=%4d: 8x8 %s |sum| %.4e   oracle restatement bit-equal: %s" % (C, a8.shape, np.abs(a8).sum(), same))
    save("roi_align_fwd", "extracted", **out)


# ----------------------------------------------------------------------------- the two eval tails
def gen_eval_tails():
    """f1: the per-image detection loop of test_net_instance_styleD_bilinear.py:151-221 driven with the reference's own
    importable pieces (bbox_transform_inv, clip_boxes, nms_cpu -- what model.nms.nms_wrapper.nms always calls) in the loop's
    order.  The loop itself lives in a script that needs a dataset and a checkpoint, so its control flow is restated here, line
    by line; every arithmetic step is the reference's function.  Tier "direct".
    f3: lib/utils.py:584-628 ``detection_output``.  lib/utils.py cannot be imported (a module-level json.load of an absolute
    path, :34-35), so the function's own source lines are compiled from the file (ast) and run unmodified; it spells the
    float64 dtype ``np.float``, an alias numpy removed in 1.24 -- restored for the call.  Tier "extracted"."""
    import ast
    from model.rpn import bbox_transform as bt
    from model.nms.nms_cpu import nms_cpu

    def nms(dets, thresh):                                  # nms_wrapper.py:13-20
        if dets.shape[0] == 0:
            return []
        return nms_cpu(dets.cpu(), thresh)

    out = {}
    for tag, (C, R, thresh, scale, agnostic) in {"c16": (16, 300, 0.0, 1.6, False), "c8_t05": (8, 120, 0.05, 1.0, False),
                                                 "c16_cag": (16, 300, 0.0, 1.25, True)}.items():
        rng = np.random.default_rng(900 + C + R)
        im_h, im_w = 600.0, 1000.0
        xy = rng.uniform(0, 1, (R, 2)) * [im_w - 120, im_h - 120]
        wh = rng.uniform(16, 300, (R, 2))
        rois = np.concatenate([np.zeros((R, 1)), xy, np.minimum(xy + wh, [im_w - 1, im_h - 1])], 1).astype(np.float32)
        logits = (rng.standard_normal((R, C)) * 2).astype(np.float32)
        prob = torch.softmax(torch.from_numpy(logits), 1).numpy()
        assert len(np.unique(prob)) == prob.size                         # tie-free scores
        pred = (rng.standard_normal((R, 4 if agnostic else 4 * C)) * 0.5).astype(np.float32)
        stds, means = (0.1, 0.1, 0.2, 0.2), (0.0, 0.0, 0.0, 0.0)         # cfg.TRAIN.BBOX_NORMALIZE_STDS / _MEANS
        # ---- :151-171
        scores = torch.from_numpy(prob).view(1, R, C)
        boxes = torch.from_numpy(rois).view(1, R, 5)[:, :, 1:5]
        im_info = torch.tensor([[im_h, im_w, scale]])
        box_deltas = torch.from_numpy(pred).view(1, R, -1)
        box_deltas = box_deltas.view(-1, 4) * torch.FloatTensor(stds) + torch.FloatTensor(means)
        box_deltas = box_deltas.view(1, -1, 4) if agnostic else box_deltas.view(1, -1, 4 * C)
        pred_boxes = bt.bbox_transform_inv(boxes, box_deltas, 1)
        pred_boxes = bt.clip_boxes(pred_boxes, im_info, 1)
        pred_boxes /= im_info[0][2].item()
        scores = scores.squeeze()
        pred_boxes = pred_boxes.squeeze()
        # ---- :181-207
        all_boxes = [np.zeros((0, 5), np.float32)]
        for j in range(1, C):
            inds = torch.nonzero(scores[:, j] > thresh).view(-1)
            if inds.numel() > 0:
                cls_scores = scores[:, j][inds]
                _, order = torch.sort(cls_scores, 0, True)
                cls_boxes = pred_boxes[inds, :] if agnostic else pred_boxes[inds][:, j * 4:(j + 1) * 4]
                cls_dets = torch.cat((cls_boxes, cls_scores.unsqueeze(1)), 1)
                cls_dets = cls_dets[order]
                keep = nms(cls_dets, 0.3)                   # cfg.TEST.NMS
                cls_dets = cls_dets[keep.view(-1).long()]
                all_boxes.append(cls_dets.cpu().numpy())
            else:
                all_boxes.append(np.zeros((0, 5), np.float32))
        # ---- :214-221
        max_per_image = 100
        image_scores = np.hstack([all_boxes[j][:, -1] for j in range(1, C)])
        before = len(image_scores)
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, C):
                keepj = np.where(all_boxes[j][:, -1] >= image_thresh)[0]
                all_boxes[j] = all_boxes[j][keepj, :]
        out[tag + "_rois"], out[tag + "_prob"], out[tag + "_pred"] = rois, prob, pred
        out[tag + "_args"] = np.array([im_h, im_w, scale, float(agnostic), thresh, 0.3, max_per_image], np.float64)
        out[tag + "_count"] = np.array([a.shape[0] for a in all_boxes], np.int32)
        out[tag + "_dets"] = np.concatenate(all_boxes, 0).astype(np.float32)
        print("    detection loop %-8s %3d rois x %2d classes: %4d detections after NMS, %3d kept" % (
            tag, R, C, before, out[tag + "_dets"].shape[0]))
    save("det_postprocess", "direct", **out)

    # ---- f3: detection_output, compiled from its own lines of lib/utils.py
    path = os.path.join(REF, "lib", "utils.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "detection_output")
    assert (fn.lineno, fn.end_lineno) == (584, 627), (fn.lineno, fn.end_lineno)      # :628 is the blank line after the return
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    detection_output = ns["detection_output"]
    had = hasattr(np, "float")
    if not had:
        np.float = float                                    # the alias numpy 1.24 removed (the function's dtype spelling)
    try:
        out = {}
        for tag, (nb, nrel) in {"b9": (9, 26), "b5": (5, 26), "b16": (16, 62), "b1": (1, 26)}.items():
            rng = np.random.default_rng(77 + nb)
            xy = rng.uniform(0, 400, (nb, 2))
            bboxes = np.concatenate([xy, xy + rng.uniform(20, 200, (nb, 2))], 1)
            classes = rng.integers(1, 16, nb)
            confs = rng.uniform(0.05, 1.0, nb).astype(np.float32)
            pairs = [(i, j) for i in range(nb) for j in range(nb) if i != j]        # faster_rcnn_SGG_emb.py:597-606
            ixs, ixo = np.array([p[0] for p in pairs], np.int64), np.array([p[1] for p in pairs], np.int64)
            rel = torch.softmax(torch.from_numpy(rng.standard_normal((max(len(pairs), 1), nrel)).astype(np.float32) * 2), 1)
            vrd_data = {"bboxes": bboxes, "classes": classes, "scores": confs, "ixs": ixs, "ixo": ixo, "rel_score": rel.clone(),
                        "rel_so_prior": None}
            res = detection_output(vrd_data)
            out[tag + "_bboxes"], out[tag + "_classes"], out[tag + "_scores"] = bboxes, classes, confs
            out[tag + "_ixs"], out[tag + "_ixo"], out[tag + "_rel_score"] = ixs, ixo, rel.numpy()
            if res[0] is None:
                out[tag + "_none"] = np.array(1)
                continue
            rlp, conf, sub, obj, idx = res
            assert len(np.unique(conf)) == conf.size        # tie-free ranking
            out[tag + "_rlp"], out[tag + "_conf"], out[tag + "_sub"], out[tag + "_obj"], out[tag + "_idx"] = rlp, conf, sub, obj, idx
            print("    detection_output %-4s %2d boxes, %3d pairs x %2d predicates -> %3d triplets, best %.4f" % (
                tag, nb, len(pairs), nrel, conf.size, conf[0]))
    finally:
        if not had:
            del np.float
    save("detection_output", "extracted", **out)


# ----------------------------------------------------------------------------- placeholders
class _AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def install_placeholders():
    ed = types.ModuleType("easydict")
    ed.EasyDict = _AttrDict
    sys.modules["easydict"] = ed
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
    sys.modules["cv2"] = types.ModuleType("cv2")
    for ext in ("roi_align", "roi_pooling", "roi_crop", "nms"):
        m = types.ModuleType("model.%s._ext" % ext)
        setattr(m, ext, types.ModuleType(ext))
        sys.modules["model.%s._ext" % ext] = m
        sys.modules["model.%s._ext.%s" % (ext, ext)] = getattr(m, ext)

    import yaml
    from model.utils import config
    with open(os.path.join(REF, "cfgs", "res101.yml")) as f:
        config._merge_a_into_b(_AttrDict(yaml.safe_load(f)), config.cfg)
    config.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]",
                          "MAX_NUM_GT_BOXES", "30"])          # parser_func.py:198-199
    return config.cfg


def _rpn_inputs(seed, B, H=38, W=63):
    """Synthetic rpn_cls_prob / rpn_bbox_pred maps with tie-free fg scores."""
    rng = np.random.default_rng(seed)
    n = B * 9 * H * W
    fg = (rng.permutation(n).astype(np.float32) + 1.0) / np.float32(n + 2)     # distinct, in (0,1)
    fg = fg.reshape(B, 9, H, W)
    prob = np.concatenate([1.0 - fg, fg], 1).astype(np.float32)
    deltas = (rng.standard_normal((B, 36, H, W)) * 0.25).astype(np.float32)
    return prob, deltas


def gen_rpn_layers(cfg):
    from model.rpn.proposal_layer import _ProposalLayer
    from model.rpn.anchor_target_layer import _AnchorTargetLayer
    from model.rpn.proposal_target_layer_cascade import _ProposalTargetLayer

    layer = _ProposalLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    out = {}
    for B in (1, 2):
        prob, deltas = _rpn_inputs(200 + B, B)
        info = np.array([[600, 1000, 1.0], [600, 1000, 1.0]], np.float32)[:B]
        for mode, key, target, post in (("train", "TRAIN", False, None), ("test", "TEST", False, None),
                                        ("target", "TRAIN", True, 32)):
            if post is not None:
                cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = post
            rois = layer((torch.from_numpy(prob), torch.from_numpy(deltas), torch.from_numpy(info), key),
                         target=target)
            out["rois_B%d_%s" % (B, mode)] = rois.numpy()
    cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 128
    save("proposal_layer", "placeholders", **out)

    # anchor targets: np.random call order is part of the contract (seed = cfg.RNG_SEED = 3)
    atl = _AnchorTargetLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    out = {}
    for B in (1, 2):
        gt, nb = syn.gt_boxes(300 + B, B, 8)
        info = torch.tensor([[600, 1000, 1.0]] * B)
        np.random.seed(3)
        res = atl((torch.zeros(B, 18, 38, 63), torch.from_numpy(gt), info, torch.from_numpy(nb)))
        for name, t in zip(("labels", "targets", "inw", "outw"), res):
            out["B%d_%s" % (B, name)] = t.numpy()
    save("anchor_target", "placeholders", **out)

    ptl = _ProposalTargetLayer(16)
    out = {}
    for B, R in ((1, 128), (2, 32)):
        cfg.TRAIN.BATCH_SIZE = R
        gt, nb = syn.gt_boxes(400 + B, B, 8)
        rois = np.zeros((B, 2000, 5), np.float32)
        for b in range(B):
            rois[b, :, 0] = b
            rois[b, :, 1:] = syn.boxes(410 + b, 2000, min_side=24, max_side=380)
            # make some proposals near gt so that fg exists
            jit = np.random.default_rng(420 + b).normal(0, 8, (64, 4)).astype(np.float32)
            rois[b, :64, 1:] = np.clip(gt[b, np.arange(64) % 8, :4] + jit, 0, [999, 599, 999, 599])
        np.random.seed(3)
        res = ptl(torch.from_numpy(rois), torch.from_numpy(gt), torch.from_numpy(nb))
        for name, t in zip(("rois", "labels", "targets", "inw", "outw"), res):
            out["B%d_R%d_%s" % (B, R, name)] = t.numpy()
    cfg.TRAIN.BATCH_SIZE = 128
    save("proposal_target", "placeholders", **out)


def _load(module, params, prefix):
    sd = {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    left = [k for k in missing.missing_keys if "num_batches_tracked" not in k]
    assert not left, left
    return module


def gen_nets(cfg):
    import model.faster_rcnn.resnet_instance_styleD_bilinear as R
    from model.rpn.rpn import _RPN
    from model.utils.net_utils import _smooth_l1_loss

    # --- discriminators: outputs + grads through the GRL
    p = syn.netd_params(12)
    dp = _load(R.netD_pixel(context=True), p, "netD_pixel.")
    ds = _load(R.netD_style(context=True), p, "netD_style.")
    rng = np.random.default_rng(500)
    x = torch.from_numpy(rng.standard_normal((6, 1024, 7, 7), dtype=np.float32)).requires_grad_()
    d, feat = dp(x, 0.1)
    loss = 0.5 * torch.mean(d ** 2) + feat.sum() * 1e-3
    loss.backward()
    out = dict(pix_d=d.detach().numpy(), pix_feat=feat.detach().numpy(), pix_gx=x.grad.numpy()[:, ::16].copy(),
               pix_gw1=dp.conv1.weight.grad.numpy()[:8], pix_gw3=dp.conv3.weight.grad.numpy())
    y = torch.from_numpy(rng.standard_normal((2, 512, 19, 32), dtype=np.float32)).requires_grad_()
    d, feat = ds(y, 0.01)
    loss = 0.5 * torch.mean((1 - d) ** 2)
    loss.backward()
    out.update(sty_d=d.detach().numpy(), sty_feat=feat.detach().numpy(), sty_gx=y.grad.numpy()[:, ::16].copy(),
               sty_gw1=ds.fc_1.weight.grad.numpy()[:16], sty_gb2=ds.fc_2.bias.grad.numpy(),
               sty_gfc1=ds.fc1.weight.grad.numpy())
    save("discriminators", "placeholders", **out)

    # --- backbone pieces on reduced spatial size (ResNet-101 blocks, frozen BN, eval)
    bp = syn.backbone_params(0, 101)
    net = R.resnet101()
    sd = {}
    names = {"RCNN_base.0": "conv1", "RCNN_base.1": "bn1", "RCNN_base.4": "layer1", "RCNN_base.5": "layer2",
             "RCNN_base.6": "layer3", "RCNN_top.0": "layer4"}
    for k, v in bp.items():
        for a, b in names.items():
            if k.startswith(a + "."):
                sd[b + k[len(a):]] = v
    miss = net.load_state_dict(sd, strict=False)
    assert not miss.unexpected_keys
    assert all(("fc." in k) or ("num_batches" in k) for k in miss.missing_keys), miss.missing_keys
    net.eval()
    base = torch.nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool, net.layer1, net.layer2, net.layer3)
    im, _ = syn.frames(600, 2, 97, 131)
    with torch.no_grad():
        x = torch.from_numpy(im)
        taps = {}
        for i, m in enumerate(base):
            x = m(x)
            if i in (3, 4, 5, 6):
                taps["after_%d" % i] = x.numpy()
        rng = np.random.default_rng(601)
        pool5 = torch.from_numpy(rng.standard_normal((3, 1024, 7, 7), dtype=np.float32))
        taps["head_to_tail"] = net.layer4(pool5).mean(3).mean(2).numpy()
    # keep the fixture small: layer taps as (sum, abs-sum, strided sample) + the final map in full
    out = {}
    for k, v in taps.items():
        out[k + "_shape"] = np.array(v.shape)
        out[k + "_sum"] = np.array(v.astype(np.float64).sum())
        out[k + "_abs"] = np.array(np.abs(v.astype(np.float64)).sum())
        out[k + "_sample"] = v.reshape(-1)[::53].copy()
    out["after_6_full"] = taps["after_6"]
    out["head_to_tail_full"] = taps["head_to_tail"]
    save("backbone_small", "placeholders", **out)

    # --- RPN head + train-mode losses (B=2)
    rp = syn.rpn_params(10)
    rpn = _load(_RPN(1024), rp, "RCNN_rpn.")
    rpn.train()
    rng = np.random.default_rng(700)
    feat = torch.from_numpy(np.abs(rng.standard_normal((2, 1024, 38, 63), dtype=np.float32)))
    gt, nb = syn.gt_boxes(701, 2, 8)
    info = torch.tensor([[600, 1000, 1.0]] * 2)
    cfg.TRAIN.RPN_POST_NMS_TOP_N = 2000
    np.random.seed(3)
    rois, lc, lb = rpn(feat, info, torch.from_numpy(gt), torch.from_numpy(nb))
    save("rpn_train", "placeholders", rois=rois.numpy(), loss_cls=lc.detach().numpy(),
         loss_box=lb.detach().numpy())

    # --- smooth L1
    rng = np.random.default_rng(710)
    a, b = (torch.from_numpy(rng.standard_normal((64, 4), dtype=np.float32)) for _ in range(2))
    iw = torch.from_numpy((rng.random((64, 4)) > 0.5).astype(np.float32))
    save("smooth_l1", "placeholders",
         s1=_smooth_l1_loss(a, b, iw, iw).numpy(),
         s3=_smooth_l1_loss(a.view(2, 8, 4, 4), b.view(2, 8, 4, 4), iw.view(2, 8, 4, 4), iw.view(2, 8, 4, 4) * 0.01,
                            sigma=3, dim=[1, 2, 3]).numpy())


def gen_full_frame(cfg):
    """the reference ResNet on FULL 600x1000 frames, pinned by shape + sums + a strided sample
    (the real 150x250 / 75x125 / 38x63 tile-padding paths of the implicit-GEMM and Winograd kernels).
    res101 on the two bench frames (configs[1]); res50 on one frame (configs[0], cfgs/res50.yml plumbing)."""
    import model.faster_rcnn.resnet_instance_styleD_bilinear as R
    names = {"RCNN_base.0": "conv1", "RCNN_base.1": "bn1", "RCNN_base.4": "layer1", "RCNN_base.5": "layer2",
             "RCNN_base.6": "layer3", "RCNN_top.0": "layer4"}
    out = {}
    for layers, ctor, nfr, seed in ((101, R.resnet101, 2, 1), (50, R.resnet50, 1, 0)):
        bp = syn.backbone_params(0, layers)
        net = ctor()
        sd = {}
        for k, v in bp.items():
            for a, b in names.items():
                if k.startswith(a + "."):
                    sd[b + k[len(a):]] = v
        miss = net.load_state_dict(sd, strict=False)
        assert not miss.unexpected_keys
        assert all(("fc." in k) or ("num_batches" in k) for k in miss.missing_keys), miss.missing_keys
        net.eval()
        im, _ = syn.frames(seed, nfr, 600, 1000)
        with torch.no_grad():
            x = net.maxpool(net.relu(net.bn1(net.conv1(torch.from_numpy(im)))))
            f1 = net.layer2(net.layer1(x))          # base_feat1: the style tap (resnet_instance...:412-420)
            f = net.layer3(f1)                      # base_feat
        for key, v in (("feat1", f1.numpy()), ("feat", f.numpy())):
            tag = "r%d_%s" % (layers, key)
            out[tag + "_shape"] = np.array(v.shape)
            out[tag + "_sum"] = np.array(v.astype(np.float64).sum())
            out[tag + "_abs"] = np.array(np.abs(v.astype(np.float64)).sum())
            out[tag + "_sample"] = v.reshape(-1)[::251].copy()
        print("   res%d: feat %s |sum| %.4e" % (layers, tuple(f.shape), float(out["r%d_feat_abs" % layers])))
    save("backbone_full_frame", "placeholders", **out)


def gen_context(cfg):
    """the reference ``_fasterRCNN`` (instance_styleD) with ic = gc = True run as shipped --
    context vectors of both discriminators concatenated in front of the layer4 feature
    (faster_rcnn_instance_styleD_bilinear.py:62-67,122-148) -- with the two pieces the reference cannot run here
    supplied by the harness: the RPN (fixed proposals, so that no near-tied score decides the sample) and RoIAlignAvg
    (legacy autograd.Function over an unbuildable extension -> the reference's compiled ROIAlignForwardCpu, oracle/build_ref.py).  Forward only, training mode."""
    import model.faster_rcnn.resnet_instance_styleD_bilinear as R
    n_cls = 16
    cfg.TRAIN.BATCH_SIZE = 32
    try:
        net = R.resnet(tuple(range(n_cls)), 50, pretrained=False, class_agnostic=False, ic=True, gc=True)
        net.create_architecture()
        p = {}
        p.update(syn.backbone_params(0, 50, top=True))
        p.update(syn.det_head_params(11, n_cls, feat_d=2048 + 512 + 128))
        p.update(syn.netd_params(12))
        miss = net.load_state_dict(p, strict=False)
        assert not miss.unexpected_keys, miss.unexpected_keys
        assert all(k.startswith("RCNN_rpn.") or "num_batches" in k for k in miss.missing_keys), miss.missing_keys
        im, info, gt, nb, rois = syn.context_inputs()

        class FixedRPN(torch.nn.Module):
            def forward(self, base_feat, im_info, gt_boxes, num_boxes, target=False):
                return torch.from_numpy(rois), torch.zeros(1), torch.zeros(1)

        class HarnessAlign(torch.nn.Module):
            def forward(self, feat, r):
                from oracle import build_ref          # the reference's own compiled forward (bit-equal to oracle.cops)
                x = torch.from_numpy(build_ref.roi_align_fwd(feat.detach().numpy(), r.detach().numpy(), 8, 8, 1.0 / 16.0))
                return torch.nn.functional.avg_pool2d(x, kernel_size=2, stride=1)

        net.RCNN_rpn = FixedRPN()
        net.RCNN_roi_align = HarnessAlign()
        net.train()
        np.random.seed(3)
        with torch.no_grad():
            out = net(torch.from_numpy(im), torch.from_numpy(info), torch.from_numpy(gt), torch.from_numpy(nb),
                      target=False, eta=0.1, eta_style=0.001)
        r, cls_prob, bbox_pred, _, _, l_cls, l_box, label, d_inst, d_sty = out
        save("context_ic_gc", "placeholders", rois=r.numpy(), labels=label.numpy(), cls_prob=cls_prob.numpy(),
             bbox_pred=bbox_pred.numpy(), loss_cls=l_cls.numpy(), loss_box=l_box.numpy(), d_instance=d_inst.numpy(),
             d_style=d_sty.numpy())
    finally:
        cfg.TRAIN.BATCH_SIZE = 128


def gen_step(cfg):
    """the reference ``_fasterRCNN`` (instance_styleD, ic = gc = False) run as shipped for ONE D+G step's
    forward passes -- source (faster_rcnn_instance_styleD_bilinear.py:47-182, target=False) then target (target=True) -- and
    the eight scalars the loop assembles from them (trainval_net_instance_styleD_bilinear.py:276-296).  Its own backbone, own
    ``_RPN`` (proposal layer with ``nms_cpu``, anchor targets), own ``_ProposalTargetLayer`` under ``np.random.seed(3)``, own
    discriminators and heads.  The one piece it cannot run here is ``RoIAlignFunction`` (a torch-0.4 legacy Function over an
    extension that needs TH); the harness's ``RCNN_roi_align`` does what ``RoIAlignAvg.forward`` does
    (roi_align/modules/roi_align.py:26-29) with the reference's own compiled ``ROIAlignForwardCpu`` (oracle/build_ref.py)
    in the Function's place: align to 8x8, ``avg_pool2d(x, kernel_size=2, stride=1)``.  Forward only, training mode.
    Two cases: res101 on 2+2 frames of 320x480, and res101 on 1+1 full 600x1000 frames; 32 ROIs per frame (configs[2])."""
    import model.faster_rcnn.resnet_instance_styleD_bilinear as R
    from oracle import build_ref
    build_ref.build()
    n_cls = 16

    class RefAlignAvg(torch.nn.Module):
        def forward(self, features, rois):
            x = torch.from_numpy(build_ref.roi_align_fwd(features.detach().numpy(), rois.detach().numpy(), 7 + 1, 7 + 1, 1.0 / 16.0))
            return torch.nn.functional.avg_pool2d(x, kernel_size=2, stride=1)

    cfg.TRAIN.BATCH_SIZE = 32
    cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 32
    cfg.TRAIN.RPN_POST_NMS_TOP_N = 2000
    out = {}
    try:
        for tag, B, H, W in (("small", 2, 320, 480), ("full", 1, 600, 1000)):
            net = R.resnet(tuple(range(n_cls)), 101, pretrained=False, class_agnostic=False)
            net.create_architecture()
            p = syn.instance_styled_step_params()
            miss = net.load_state_dict(p, strict=False)
            assert not miss.unexpected_keys, miss.unexpected_keys
            assert all("num_batches" in k for k in miss.missing_keys), miss.missing_keys
            net.RCNN_roi_align = RefAlignAvg()
            net.train()
            im, info, gt, nb, im_t, info_t = syn.instance_styled_step_inputs(B, H, W, n_cls)
            stash = []
            net.RCNN_rpn.register_forward_hook(lambda m, i, o: stash.append(o[0].detach().numpy().copy()))
            np.random.seed(3)
            with torch.no_grad():
                (rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_box, RCNN_loss_cls, RCNN_loss_bbox, rois_label, d_inst,
                 d_sty) = net(torch.from_numpy(im), torch.from_numpy(info), torch.from_numpy(gt), torch.from_numpy(nb),
                              target=False, eta=0.1, eta_style=0.001)
                d_inst_t, d_sty_t = net(torch.from_numpy(im_t), torch.from_numpy(info_t), torch.zeros(B, 1, 5), torch.zeros(B),
                                        target=True, eta=0.1, eta_style=0.001)
            losses = np.array([rpn_loss_cls.mean(), rpn_loss_box.mean(), RCNN_loss_cls.mean(), RCNN_loss_bbox.mean(),
                               0.5 * torch.mean(d_inst ** 2), 0.5 * torch.mean(d_sty ** 2),
                               0.5 * torch.mean((1 - d_inst_t) ** 2), 0.5 * torch.mean((1 - d_sty_t) ** 2)], np.float64)
            print("   %s: " % tag + " ".join("%.6f" % v for v in losses))
            out.update({tag + "_losses": losses, tag + "_rpn_rois_src": stash[0], tag + "_rpn_rois_tgt": stash[1],
                        tag + "_rois": rois.numpy(), tag + "_labels": rois_label.numpy(), tag + "_cls_prob": cls_prob.numpy(),
                        tag + "_bbox_pred": bbox_pred.numpy(), tag + "_d_instance": d_inst.numpy(), tag + "_d_style": d_sty.numpy(),
                        tag + "_d_instance_t": d_inst_t.numpy(), tag + "_d_style_t": d_sty_t.numpy()})
        out["loss_names"] = np.array(["rpn_loss_cls", "rpn_loss_box", "RCNN_loss_cls", "RCNN_loss_bbox", "dloss_s_p",
                                      "dloss_s_style", "dloss_t_p", "dloss_t_style"])
        save("instance_styled_step", "placeholders+extracted", **out)
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
        cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 128


def gen_vrd(cfg):
    """vrd.forward logits / BCE loss / grads with eval-mode dropout."""
    import pickle
    import tempfile
    C = types.ModuleType("model._C")

    def roi_pool_forward(inp, rois, scale, ph, pw):
        out, arg = cops.roi_pool_fwd(inp.detach().numpy(), rois.detach().numpy(), ph, pw, scale)
        return torch.from_numpy(out), torch.from_numpy(arg)
    C.roi_pool_forward = roi_pool_forward
    C.nms = None            # bound at import by roi_layers/nms.py:5, never called on this path
    import model
    model._C = C
    sys.modules["model._C"] = C
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # faster_rcnn_SGG_emb.py imports at module level things this path never calls
    import model.faster_rcnn.resnet_SGG_emb as S
    import torch.nn.functional as F

    n_rel, n_cls = 62, 16
    tmp = tempfile.mkdtemp()
    paths = {}
    for k in ("so_prior", "gt_s", "gt_t"):
        paths[k] = os.path.join(tmp, k + ".pkl")
        with open(paths[k], "wb") as f:
            pickle.dump({} if k != "so_prior" else [[0.0]], f)
    args = argparse.Namespace(num_relations=n_rel, num_classes=n_cls, emb_dim=300, use_obj_visual=True,
                              spatial_type=2, source_so_prior_path=paths["so_prior"],
                              source_gt_rels_path=paths["gt_s"], target_gt_rels_path=paths["gt_t"])
    prd = syn.word_vectors(21, n_rel)
    obj = syn.word_vectors(22, n_cls)
    head = S.vrd(args, obj, prd)
    p = syn.vrd_params(13)
    _load(head, p, "vrd.")
    head.eval()            # dropout off; eval also applies the softmax -> undo by calling in train w/o dropout
    head.training = True
    F_dropout = F.dropout
    S.F.dropout = lambda x, training=True: x
    try:
        from oracle import nets
        anno = syn.relation_annotation(31, 8, 8, n_rel, n_cls)
        ih, iw = 600.0, 1000.0
        boxes, rel_boxes, spatial, labels, ixs, ixo = nets.build_pairs(anno["boxes"], anno["rels"], 1.0, ih, iw, n_rel)
        rng = np.random.default_rng(32)
        fmap = np.abs(rng.standard_normal((1, 1024, 38, 63), dtype=np.float32))
        # pair builder pinned against the reference's own helper methods
        ub = np.array([head._getUnionBBox(np.array(anno["boxes"][s]), np.array(anno["boxes"][o]), ih, iw)
                       for s, o in zip(ixs, ixo)])
        dm = np.array([head._getDualMask(ih, iw, np.array(anno["boxes"][s])) for s in ixs])
        classes = np.array(anno["box_classes"]).astype(np.float32)
        score, feat = head(fmap, boxes, rel_boxes, spatial, classes, ixs, ixo)
        loss = head.criterion(score, torch.from_numpy(labels).float())
        loss.backward()
        save("vrd_head", "placeholders", union_boxes=ub, dual_masks=dm, scores=score.detach().numpy(),
             rel_feat=feat, loss=loss.detach().numpy(),
             g_fc6_w=head.fc6.fc.weight.grad.numpy()[:4, ::97].copy(),
             g_fc6_b=head.fc6.fc.bias.grad.numpy(), g_fc7_b=head.fc7.fc.bias.grad.numpy(),
             g_fc_rel_w=head.fc_rel.fc.weight.grad.numpy(),
             g_conv0_w=head.conv_lo[0].conv.weight.grad.numpy(),
             g_sem0_b=head.prd_sem_embeddings[0].bias.grad.numpy(),
             g_fc6_w_sum=np.array(head.fc6.fc.weight.grad.numpy().astype(np.float64).sum()),
             g_fc6_w_abs=np.array(np.abs(head.fc6.fc.weight.grad.numpy().astype(np.float64)).sum()))
        # the variants of resnet_SGG_emb.py:94-123 / :166-180 (the reference's scripts only ever run the defaults: its
        # type=bool flags parse every CLI value to True -- but the class implements them, so they are pinned too)
        out = {}
        for tag, (ov, st) in {"nov_s1": (False, 1), "ov_s1": (True, 1), "nov_s2": (False, 2), "ov_s0": (True, 0)}.items():
            a2 = argparse.Namespace(**dict(vars(args), use_obj_visual=ov, spatial_type=st))
            h2 = S.vrd(a2, obj, prd)
            _load(h2, syn.vrd_params(13, use_obj_visual=ov, spatial_type=st), "vrd.")
            h2.eval()
            h2.training = True
            if st == 1:
                sp = np.array([h2._getRelativeLoc(np.array(anno["boxes"][s_]), np.array(anno["boxes"][o_])) for s_, o_ in zip(ixs, ixo)])
                out[tag + "_spatial"] = sp
            else:
                sp = spatial
            sc2, ft2 = h2(fmap, boxes, rel_boxes, sp, classes, ixs, ixo)
            l2 = h2.criterion(sc2, torch.from_numpy(labels).float())
            l2.backward()
            out[tag + "_scores"], out[tag + "_rel_feat"], out[tag + "_loss"] = sc2.detach().numpy(), ft2, l2.detach().numpy()
            out[tag + "_g_fusion_w"] = h2.fc_fusion.fc.weight.grad.numpy()[::16].copy()
            out[tag + "_g_fc7_b"] = h2.fc7.fc.bias.grad.numpy()
            if st in (1, 2):
                out[tag + "_g_lov_w"] = h2.fc_lov.fc.weight.grad.numpy()[::4].copy()
            print("    vrd variant %-7s use_obj_visual=%s spatial_type=%d: loss %.6f, fc_fusion %s" % (
                tag, ov, st, float(l2), tuple(h2.fc_fusion.fc.weight.shape)))
        save("vrd_head_variants", "placeholders", **out)
    finally:
        S.F.dropout = F_dropout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    todo = a.only.split(",") if a.only else ["direct", "roialign", "tails", "rpn", "nets", "full", "ctx", "step", "vrd"]
    if "direct" in todo:
        print("[direct imports]")
        gen_direct()
    if "roialign" in todo:
        print("[RoIAlign forward: the reference's C function compiled from its own lines]")
        gen_roi_align()
    if "tails" in todo:
        print("[eval tails: direct imports + detection_output compiled from its own lines]")
        gen_eval_tails()
    if any(t in todo for t in ("rpn", "nets", "full", "vrd", "ctx", "step")):
        print("[imports with placeholders]")
        cfg = install_placeholders()
        if "rpn" in todo:
            gen_rpn_layers(cfg)
        if "nets" in todo:
            gen_nets(cfg)
        if "full" in todo:
            gen_full_frame(cfg)
        if "ctx" in todo:
            gen_context(cfg)
        if "step" in todo:
            gen_step(cfg)
        if "vrd" in todo:
            gen_vrd(cfg)


if __name__ == "__main__":
    main()
