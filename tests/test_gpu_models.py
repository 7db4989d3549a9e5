"""Model-level parity on the GPU: the HIP path (through the reference-named modules) against the
golden vectors the reference itself produced (tools/gen_golden.py) and against the CPU oracle.
Tolerance for floating point: 1e-3 relative (BASELINE.json north_star), written per assert."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from i2vsgg_amd import synthetic as syn  # noqa: E402

DEV = "cuda:0"
REL = 1e-3


@pytest.fixture(scope="module")
def cfg():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from i2vsgg_amd.model.utils import config as c
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
    return c.cfg


def _rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _weights_close(got, want, what, l2=1e-5, peak=1e-5):
    """Two training trajectories that differ only in schedule (eager / captured, one pass / per-frame branches): the filters must
    agree in the L2 sense to ``l2`` and element-wise to ``peak`` of the largest filter value.
    Round 4 had to allow 1e-4 element-wise: the small FCs of the relation head accumulated their split-K partials with fp32
    atomics, so a pre-activation within rounding of zero could land on either side of its ReLU from one run to the next (seen
    once in five suite runs: 1.5e-5).  Round 5 removed the cause -- every split reduction of the step is summed in a fixed
    order (csrc/conv.hip: split-K finish for any split count, ordered filter-gradient and bias-column sums;
    test_sgg_step_is_bit_reproducible) -- and the bound is back at 1e-5.  Every deviation above 1e-6 is still logged with its
    shape (how many filter rows / columns carry it) to gpurun_out/weights_deviation.log, pass or fail."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    d = np.abs(got - want)
    top = max(np.abs(want).max(), 1e-30)
    r_peak, r_l2 = d.max() / top, np.linalg.norm(d) / max(np.linalg.norm(want), 1e-30)
    if r_peak > 1e-6:
        d2 = d.reshape(d.shape[0], -1)
        rows, cols = int((d2.max(1) > 1e-6 * top).sum()), int((d2.max(0) > 1e-6 * top).sum())
        line = "%s: peak %.3e, L2 %.3e, %d of %d rows and %d of %d columns above 1e-6" % (
            what, r_peak, r_l2, rows, d2.shape[0], cols, d2.shape[1])
        print(line)
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            with open(os.path.join(root, "gpurun_out", "weights_deviation.log"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
    assert r_l2 < l2 and r_peak < peak, (what, r_l2, r_peak)


def _load(module, params, prefix=""):
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    sd = {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}
    r = load_reference_state(module, sd, strict=False)
    assert not r.unexpected_keys, r.unexpected_keys
    return module


def test_backbone_and_layer4_vs_reference_golden(cfg, gold):
    from i2vsgg_amd.model.faster_rcnn.layers import C4Base, make_layer
    g = gold("backbone_small")
    p = syn.backbone_params(0, 101)
    base = _load(C4Base((3, 4, 23)), p, "RCNN_base.").to(DEV)
    left = [k for k in base.state_dict() if ("RCNN_base." + k) not in p and "num_batches" not in k]
    assert not left, left                      # every reference key is consumed
    im, _ = syn.frames(600, 2, 97, 131)
    with torch.no_grad():
        feat, feat1 = base(torch.from_numpy(im).to(DEV), tap=True)
    assert tuple(feat.shape) == tuple(g["after_6_shape"]) and tuple(feat1.shape) == tuple(g["after_5_shape"])
    assert _rel_err(feat.cpu().numpy(), g["after_6_full"]) < REL
    assert _rel_err(feat1.contiguous().cpu().numpy().reshape(-1)[::53], g["after_5_sample"]) < REL
    layer4, _ = make_layer(1024, 512, 3, 2)
    top = _load(torch.nn.Sequential(layer4), p, "RCNN_top.").to(DEV)
    pool5 = torch.from_numpy(np.random.default_rng(601).standard_normal((3, 1024, 7, 7), dtype=np.float32)).to(DEV)
    with torch.no_grad():
        h2t = top(pool5).mean(3).mean(2)
    assert _rel_err(h2t.cpu().numpy(), g["head_to_tail_full"]) < REL


@pytest.mark.parametrize("layers,blocks,nfr,seed", [(101, (3, 4, 23), 2, 1), (50, (3, 4, 6), 1, 0)])
def test_backbone_full_frame_vs_reference_golden(cfg, gold, layers, blocks, nfr, seed):
    """the C4 trunk on FULL 600x1000 frames (the 150x250 / 75x125 / 38x63 grids with their real
    tile padding, Winograd F(4x4,3x3) on the frozen 3x3 layers) against the reference ResNet run on the same frames
    (tools/gen_golden.py gen_full_frame): shape, sum, abs-sum and a strided sample of base_feat and of the style tap.
    res101 on the two configs[1] frames; res50 on the configs[0] frame."""
    from i2vsgg_amd.model.faster_rcnn.layers import C4Base
    g = gold("backbone_full_frame")
    p = syn.backbone_params(0, layers)
    base = _load(C4Base(blocks), p, "RCNN_base.").to(DEV)
    im, _ = syn.frames(seed, nfr, 600, 1000)
    with torch.no_grad():
        feat, feat1 = base(torch.from_numpy(im).to(DEV), tap=True)
    for key, t in (("feat", feat), ("feat1", feat1)):
        tag = "r%d_%s" % (layers, key)
        v = t.contiguous().cpu().numpy()                         # logical NCHW order, as the reference stores it
        assert tuple(v.shape) == tuple(g[tag + "_shape"]), tag
        assert _rel_err(v.reshape(-1)[::251], g[tag + "_sample"]) < REL, tag
        v64 = v.astype(np.float64)
        assert abs(v64.sum() - float(g[tag + "_sum"])) < REL * float(g[tag + "_abs"]), tag
        assert abs(np.abs(v64).sum() - float(g[tag + "_abs"])) < REL * float(g[tag + "_abs"]), tag


def test_context_ic_gc_vs_reference_golden(cfg, gold):
    """``_fasterRCNN`` with ic = gc = True (context vectors of both discriminators concatenated
    in front of the layer4 feature) against the reference model run with the same parameters, the same fixed
    proposals and the same np.random stream (tools/gen_golden.py gen_context)."""
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    g = gold("context_ic_gc")
    n_cls = 16
    cfg.TRAIN.BATCH_SIZE = 32
    try:
        net = resnet(tuple(range(n_cls)), 50, ic=True, gc=True)
        net.create_architecture()
        p = {}
        p.update(syn.backbone_params(0, 50, top=True))
        p.update(syn.det_head_params(11, n_cls, feat_d=2048 + 512 + 128))
        p.update(syn.netd_params(12))
        r = load_reference_state(net, p, strict=False)
        assert not r.unexpected_keys, r.unexpected_keys
        assert all(k.startswith("RCNN_rpn.") or "num_batches" in k for k in r.missing_keys), r.missing_keys
        net.to(DEV).train()
        im, info, gt, nb, rois = syn.context_inputs()
        rois_d = torch.from_numpy(rois).to(DEV)

        class FixedRPN(torch.nn.Module):
            def forward(self, base_feat, im_info, gt_boxes, num_boxes, target=False):
                z = torch.zeros(1, device=DEV)
                return rois_d, z, z

        net.RCNN_rpn = FixedRPN()
        np.random.seed(3)
        with torch.no_grad():
            out = net(torch.from_numpy(im).to(DEV), torch.from_numpy(info).to(DEV), torch.from_numpy(gt).to(DEV),
                      torch.from_numpy(nb).to(DEV), target=False, eta=0.1, eta_style=0.001)
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
    rr, cls_prob, bbox_pred, _, _, l_cls, l_box, label, d_inst, d_sty = out
    assert np.array_equal(rr.cpu().numpy(), g["rois"]) and np.array_equal(label.cpu().numpy(), g["labels"])
    assert _rel_err(cls_prob.cpu().numpy(), g["cls_prob"]) < REL
    assert _rel_err(bbox_pred.cpu().numpy(), g["bbox_pred"]) < REL
    assert _rel_err(l_cls.cpu().numpy(), g["loss_cls"]) < REL and _rel_err(l_box.cpu().numpy(), g["loss_box"]) < REL
    assert _rel_err(d_inst.cpu().numpy(), g["d_instance"]) < REL
    assert _rel_err(d_sty.cpu().numpy(), g["d_style"]) < REL


def test_discriminators_vs_reference_golden(cfg, gold):
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import netD_pixel, netD_style
    g = gold("discriminators")
    p = syn.netd_params(12)
    dp = _load(netD_pixel(context=True), p, "netD_pixel.").to(DEV)
    ds = _load(netD_style(context=True), p, "netD_style.").to(DEV)
    rng = np.random.default_rng(500)
    x = torch.from_numpy(rng.standard_normal((6, 1024, 7, 7), dtype=np.float32)).to(DEV).requires_grad_()
    d, feat = dp(x, 0.1)
    (0.5 * torch.mean(d ** 2) + feat.sum() * 1e-3).backward()
    assert _rel_err(d.detach().cpu().numpy(), g["pix_d"]) < REL
    assert _rel_err(feat.detach().cpu().numpy(), g["pix_feat"]) < REL
    assert _rel_err(x.grad.cpu().numpy()[:, ::16], g["pix_gx"]) < REL          # includes the GRL sign / scale
    assert _rel_err(dp.conv1.weight.grad.cpu().numpy()[:8], g["pix_gw1"]) < REL
    assert _rel_err(dp.conv3.weight.grad.cpu().numpy(), g["pix_gw3"]) < REL
    y = torch.from_numpy(rng.standard_normal((2, 512, 19, 32), dtype=np.float32)).to(DEV).requires_grad_()
    d, feat = ds(y, 0.01)
    (0.5 * torch.mean((1 - d) ** 2)).backward()
    assert _rel_err(d.detach().cpu().numpy(), g["sty_d"]) < REL
    assert _rel_err(feat.detach().cpu().numpy(), g["sty_feat"]) < REL
    assert _rel_err(y.grad.cpu().numpy()[:, ::16], g["sty_gx"]) < REL
    assert _rel_err(ds.fc_1.weight.grad.cpu().numpy()[:16], g["sty_gw1"]) < REL
    assert _rel_err(ds.fc_2.bias.grad.cpu().numpy(), g["sty_gb2"]) < REL
    assert _rel_err(ds.fc1.weight.grad.cpu().numpy(), g["sty_gfc1"]) < REL


@pytest.mark.parametrize("B", [1, 2])
def test_anchor_target_layer_vs_reference_golden(cfg, gold, B):
    from i2vsgg_amd.model.rpn.anchor_target_layer import _AnchorTargetLayer
    g = gold("anchor_target")
    gt, nb = syn.gt_boxes(300 + B, B, 8)
    layer = _AnchorTargetLayer(16, cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    np.random.seed(3)
    out = layer((torch.zeros(B, 18, 38, 63, device=DEV), torch.from_numpy(gt).to(DEV),
                 torch.tensor([[600, 1000, 1.0]] * B, device=DEV), torch.from_numpy(nb).to(DEV)))
    assert np.array_equal(out[0].cpu().numpy(), g["B%d_labels" % B])            # identical sampled anchors
    assert np.array_equal(out[2].cpu().numpy(), g["B%d_inw" % B])
    np.testing.assert_allclose(out[3].cpu().numpy(), g["B%d_outw" % B], rtol=1e-6)
    np.testing.assert_allclose(out[1].cpu().numpy(), g["B%d_targets" % B], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,R", [(1, 128), (2, 32)])
def test_proposal_target_layer_vs_reference_golden(cfg, gold, B, R):
    from i2vsgg_amd.model.rpn.proposal_target_layer_cascade import _ProposalTargetLayer
    g = gold("proposal_target")
    cfg.TRAIN.BATCH_SIZE = R
    try:
        gt, nb = syn.gt_boxes(400 + B, B, 8)
        rois = np.zeros((B, 2000, 5), np.float32)
        for b in range(B):
            rois[b, :, 0] = b
            rois[b, :, 1:] = syn.boxes(410 + b, 2000, min_side=24, max_side=380)
            jit = np.random.default_rng(420 + b).normal(0, 8, (64, 4)).astype(np.float32)
            rois[b, :64, 1:] = np.clip(gt[b, np.arange(64) % 8, :4] + jit, 0, [999, 599, 999, 599])
        layer = _ProposalTargetLayer(16)
        np.random.seed(3)
        out = layer(torch.from_numpy(rois).to(DEV), torch.from_numpy(gt).to(DEV), torch.from_numpy(nb).to(DEV))
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
    for name, t in zip(("rois", "labels", "targets", "inw", "outw"), out):
        ref = g["B%d_R%d_%s" % (B, R, name)]
        if name == "targets":
            np.testing.assert_allclose(t.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        else:
            assert np.array_equal(t.cpu().numpy(), ref), name


def test_proposal_layer_bit_exact_on_real_conv_head_maps(cfg):
    """Round-1 verdict: kept anchor indices were shown bit-exact only on synthetic tie-free maps.  Here the maps are the
    RPN conv head's own outputs on a realistic feature map (the HIP conv kernels, 2 frames of 38x63): the device proposal
    layer and the oracle's restatement of proposal_layer.py:49-163 + nms_cpu.py, fed the SAME probability and delta
    maps, keep the same anchors in the same order and emit bit-equal rois -- for the train (12000 -> 2000), test
    (6000 -> 300) and target (12000 -> 32) settings.  (What separates the model-level proposals from the reference's is the
    conv head's rounding, checked at 1e-3 elsewhere -- not the proposal layer.)"""
    from i2vsgg_amd import ops
    from i2vsgg_amd.model.rpn.generate_anchors import generate_anchors
    from i2vsgg_amd.model.rpn.rpn import _RPN
    from oracle import rpn as orpn
    rpn = _load(_RPN(1024), syn.rpn_params(10, std=0.02), "RCNN_rpn.").to(DEV)
    feat = torch.from_numpy(np.abs(np.random.default_rng(700).standard_normal((2, 1024, 38, 63), dtype=np.float32))).to(DEV)
    with torch.no_grad():
        cls, box = rpn.head(feat)
        B, C2, H, W = cls.shape
        A = C2 // 2
        pair = torch.softmax(torch.stack((cls[:, :A], cls[:, A:]), 0), 0)            # rpn.py:69-71
        prob = torch.cat((pair[0], pair[1]), 1).contiguous(memory_format=torch.channels_last)
    info = np.array([[600, 1000, 1.0]] * B, np.float32)
    base = torch.from_numpy(generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))).float().to(DEV)
    fg = prob[:, A:].contiguous().cpu().numpy()
    dl = box.contiguous().cpu().numpy()
    assert np.unique(fg[0]).size > 0.99 * fg[0].size                                   # realistic maps: (almost) tie-free
    for pre, post in ((12000, 2000), (6000, 300), (12000, 32)):
        rois, kept, num = ops.rpn_proposal(prob, box, torch.from_numpy(info).to(DEV), base, 16, pre, post, 0.7,
                                           is_prob=True, want_index=True)
        ref, ref_kept = orpn.proposal_layer(fg, dl, info, pre, post, 0.7)
        for b in range(B):
            n = int(num[b])
            assert n == ref_kept[b].size, (pre, post, b, n, ref_kept[b].size)
            assert np.array_equal(kept[b, :n].cpu().numpy(), ref_kept[b].astype(np.int32)), (pre, post, b)
        assert np.array_equal(rois.cpu().numpy(), ref), (pre, post)


def test_rpn_train_losses_vs_reference_golden(cfg, gold):
    from i2vsgg_amd.model.rpn.rpn import _RPN
    g = gold("rpn_train")
    rpn = _load(_RPN(1024), syn.rpn_params(10), "RCNN_rpn.").to(DEV)
    rpn.train()
    feat = torch.from_numpy(np.abs(np.random.default_rng(700).standard_normal((2, 1024, 38, 63), dtype=np.float32)))
    gt, nb = syn.gt_boxes(701, 2, 8)
    cfg.TRAIN.RPN_POST_NMS_TOP_N = 2000
    np.random.seed(3)
    rois, lc, lb = rpn(feat.to(DEV), torch.tensor([[600, 1000, 1.0]] * 2, device=DEV), torch.from_numpy(gt).to(DEV),
                       torch.from_numpy(nb).to(DEV))
    assert abs(lc.item() - float(g["loss_cls"])) / abs(float(g["loss_cls"])) < REL
    assert abs(lb.item() - float(g["loss_box"])) / abs(float(g["loss_box"])) < REL
    # proposals: same boxes up to score near-ties (softmax / conv rounding can swap neighbours):
    # compare as sets of rows per image
    got, ref = rois.cpu().numpy(), g["rois"]
    assert got.shape == ref.shape
    for b in range(2):
        a = {tuple(np.round(r, 2)) for r in got[b]}
        c = {tuple(np.round(r, 2)) for r in ref[b]}
        assert len(a & c) >= 0.98 * len(c)
    lc.backward()
    assert rpn.RPN_cls_score.weight.grad is not None and torch.isfinite(rpn.RPN_cls_score.weight.grad).all()


def _vrd_args(n_rel=62, n_cls=16):
    return argparse.Namespace(num_relations=n_rel, num_classes=n_cls, emb_dim=300, use_obj_visual=True,
                              spatial_type=2, vrd_task="pre_det")


def test_vrd_head_vs_reference_golden(cfg, gold):
    from i2vsgg_amd.model.faster_rcnn.faster_rcnn_SGG_emb import build_pair_tables, rasterize_masks
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import vrd
    g = gold("vrd_head")
    n_rel, n_cls = 62, 16
    head = vrd(_vrd_args(), syn.word_vectors(22, n_cls), syn.word_vectors(21, n_rel))
    _load(head, syn.vrd_params(13), "vrd.").to(DEV)
    head.train()
    head.dropout = False                 # fixtures use eval-mode dropout
    anno = syn.relation_annotation(31, 8, 8, n_rel, n_cls)
    gt, union, bounds, labels, ixs, ixo = build_pair_tables(anno, 1.0, 600.0, 1000.0, n_rel)
    assert np.array_equal(union, g["union_boxes"])                                # pair builder, exact
    masks = rasterize_masks(bounds, DEV)
    assert np.array_equal(masks[:, 0].cpu().numpy(), g["dual_masks"])
    fmap = np.abs(np.random.default_rng(32).standard_normal((1, 1024, 38, 63), dtype=np.float32))
    fm = torch.from_numpy(fmap).to(DEV).contiguous(memory_format=torch.channels_last)
    boxes = np.zeros((gt.shape[0], 5), np.float32)
    boxes[:, 1:] = gt
    relb = np.zeros((union.shape[0], 5), np.float32)
    relb[:, 1:] = union
    score, feat = head.forward_device(fm, torch.from_numpy(boxes).to(DEV), torch.from_numpy(relb).to(DEV), masks,
                                      torch.from_numpy(ixs).to(DEV), torch.from_numpy(ixo).to(DEV))
    loss = head.criterion(score, torch.from_numpy(labels).to(DEV))
    loss.backward()
    assert _rel_err(score.detach().cpu().numpy(), g["scores"]) < REL              # relation logits
    assert _rel_err(feat.detach().cpu().numpy(), g["rel_feat"]) < REL
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < REL
    assert _rel_err(head.fc6.fc.bias.grad.cpu().numpy(), g["g_fc6_b"]) < REL
    assert _rel_err(head.fc7.fc.bias.grad.cpu().numpy(), g["g_fc7_b"]) < REL
    assert _rel_err(head.fc_rel.fc.weight.grad.cpu().numpy(), g["g_fc_rel_w"]) < REL
    assert _rel_err(head.conv_lo[0].conv.weight.grad.cpu().numpy(), g["g_conv0_w"]) < REL
    assert _rel_err(head.prd_sem_embeddings[0].bias.grad.cpu().numpy(), g["g_sem0_b"]) < REL
    gw = head.fc6.fc.weight.grad
    assert _rel_err(gw[:4, ::97].cpu().numpy(), g["g_fc6_w"]) < REL
    assert abs(gw.double().abs().sum().item() - float(g["g_fc6_w_abs"])) / float(g["g_fc6_w_abs"]) < REL


@pytest.mark.parametrize("tag,ov,st", [("nov_s1", False, 1), ("ov_s1", True, 1), ("nov_s2", False, 2), ("ov_s0", True, 0)])
def test_vrd_head_variants_vs_reference_golden(cfg, gold, tag, ov, st):
    """The branches of ``vrd`` the reference's scripts never select but its class implements (resnet_SGG_emb.py:94-123,
    :166-180; round-3 review, missing #3): no object-visual branch (``use_obj_visual=False``), the 8-d relative-location
    feature through ``fc_lov = FC(8, 256)`` (``spatial_type=1``, ``_getRelativeLoc`` :258-264), no spatial branch; ``fc_fusion``
    as wide as the branches present.  Logits, relation feature, loss and three gradients against the reference's own outputs
    (tests/golden/vrd_head_variants.npz, tier "placeholders"), 1e-3 relative."""
    from i2vsgg_amd.model.faster_rcnn.faster_rcnn_SGG_emb import build_pair_tables, rasterize_masks
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import vrd
    g = gold("vrd_head_variants")
    n_rel, n_cls = 62, 16
    args = argparse.Namespace(num_relations=n_rel, num_classes=n_cls, emb_dim=300, use_obj_visual=ov, spatial_type=st, vrd_task="pre_det")
    head = vrd(args, syn.word_vectors(22, n_cls), syn.word_vectors(21, n_rel))
    assert hasattr(head, "fc_so") == ov and hasattr(head, "conv_lo") == (st == 2) and hasattr(head, "fc_lov") == (st in (1, 2))
    assert head.fc_fusion.fc.weight.shape[1] == 256 * (1 + int(ov) + int(st in (1, 2)))
    r = _load(head, syn.vrd_params(13, use_obj_visual=ov, spatial_type=st), "vrd.")
    head.to(DEV).train()
    head.dropout = False
    anno = syn.relation_annotation(31, 8, 8, n_rel, n_cls)
    gt, union, bounds, labels, ixs, ixo = build_pair_tables(anno, 1.0, 600.0, 1000.0, n_rel)
    if st == 1:
        sp = np.array([head._getRelativeLoc(anno["boxes"][s], anno["boxes"][o]) for s, o in zip(ixs, ixo)])
        assert np.array_equal(sp, g[tag + "_spatial"])                            # the reference's own _getRelativeLoc, bit for bit
        spatial = torch.from_numpy(sp.astype(np.float32)).to(DEV)
    else:
        spatial = rasterize_masks(bounds, DEV)
    fmap = np.abs(np.random.default_rng(32).standard_normal((1, 1024, 38, 63), dtype=np.float32))
    fm = torch.from_numpy(fmap).to(DEV).contiguous(memory_format=torch.channels_last)
    boxes = np.zeros((gt.shape[0], 5), np.float32)
    boxes[:, 1:] = gt
    relb = np.zeros((union.shape[0], 5), np.float32)
    relb[:, 1:] = union
    score, feat = head.forward_device(fm, torch.from_numpy(boxes).to(DEV), torch.from_numpy(relb).to(DEV), spatial,
                                      torch.from_numpy(ixs).to(DEV), torch.from_numpy(ixo).to(DEV))
    loss = head.criterion(score, torch.from_numpy(labels).to(DEV))
    loss.backward()
    assert _rel_err(score.detach().cpu().numpy(), g[tag + "_scores"]) < REL
    assert _rel_err(feat.detach().cpu().numpy(), g[tag + "_rel_feat"]) < REL
    assert abs(loss.item() - float(g[tag + "_loss"])) / float(g[tag + "_loss"]) < REL
    assert _rel_err(head.fc_fusion.fc.weight.grad.cpu().numpy()[::16], g[tag + "_g_fusion_w"]) < REL
    assert _rel_err(head.fc7.fc.bias.grad.cpu().numpy(), g[tag + "_g_fc7_b"]) < REL
    if st in (1, 2):
        assert _rel_err(head.fc_lov.fc.weight.grad.cpu().numpy()[::4], g[tag + "_g_lov_w"]) < REL
    # the reference-signature forward (numpy in, numpy feature out) takes the same inputs
    head.eval()
    with torch.no_grad():
        prob, f2 = head(fmap, boxes, relb, spatial.cpu().numpy(), np.asarray(anno["box_classes"], np.float32), ixs, ixo)
    assert prob.shape == (len(ixs), n_rel) and abs(float(prob.sum(1).mean()) - 1.0) < 1e-5 and f2.shape == (len(ixs), 300)


def test_sgg_emb_step_two_frames_equals_mean_of_single_frames(cfg):
    """Batch semantics: loss(B frames) == mean of the per-frame losses."""
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import resnet
    n_rel, n_cls = 62, 16
    torch.manual_seed(0)
    net = resnet(tuple(range(n_cls)), _vrd_args(), 50, obj_vecs=syn.word_vectors(22, n_cls),
                 prd_vecs=syn.word_vectors(21, n_rel))
    net.create_architecture()
    net.vrd.dropout = False
    net.vrd.source_gt_rels = {"f%d" % i: syn.relation_annotation(40 + i, 6, 5, n_rel, n_cls, 200, 320) for i in range(2)}
    net.to(DEV).train()
    im, info = syn.frames(9, 2, 200, 320)
    imd, infod = torch.from_numpy(im).to(DEV), torch.from_numpy(info).to(DEV)
    z = torch.zeros(2, 1, 5, device=DEV)
    both = net(imd, infod, z, z, ["f0", "f1"])
    one = [net(imd[i:i + 1], infod[i:i + 1], z, z, "f%d" % i) for i in range(2)]
    assert abs(both.item() - 0.5 * (one[0].item() + one[1].item())) < 1e-5 * abs(both.item())
    both.backward()
    assert net.vrd.fc6.fc.weight.grad is not None
    assert all(p.grad is None for p in net.RCNN_base.parameters())          # backbone gets no gradient (detach)


def test_instance_styled_source_and_target_step_runs_and_is_finite(cfg):
    """One D+G adversarial step (trainval_net_instance_styleD_bilinear.py:271-341) at small size."""
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    cfg.TRAIN.BATCH_SIZE = 32
    cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 32
    try:
        torch.manual_seed(0)
        np.random.seed(3)
        net = resnet(tuple(range(16)), 50)
        net.create_architecture()
        net.to(DEV).train()
        im, info = syn.frames(5, 2, 320, 480)
        gt, nb = syn.gt_boxes(6, 2, 6, im_h=320, im_w=480)
        imd, infod = torch.from_numpy(im).to(DEV), torch.from_numpy(info).to(DEV)
        out = net(imd, infod, torch.from_numpy(gt).to(DEV), torch.from_numpy(nb).to(DEV), target=False, eta=0.1,
                  eta_style=0.001)
        rois, cls_prob, bbox_pred, l1, l2, l3, l4, lab, d_inst, d_sty = out
        assert rois.shape == (2, 32, 5) and cls_prob.shape == (2, 32, 16) and d_inst.shape == (64, 1, 7, 7)
        loss = l1.mean() + l2.mean() + l3.mean() + l4.mean() + 0.5 * torch.mean(d_inst ** 2) + 0.5 * torch.mean(d_sty ** 2)
        d_it, d_st = net(imd, infod, torch.zeros(2, 1, 5, device=DEV), torch.zeros(2, device=DEV), target=True,
                         eta=0.1, eta_style=0.001)
        loss = loss + 0.5 * torch.mean((1 - d_it) ** 2) + 0.5 * torch.mean((1 - d_st) ** 2)
        loss.backward()
        assert torch.isfinite(loss)
        for name, p in net.named_parameters():
            if p.requires_grad and not name.startswith("RCNN_base.0"):
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
        assert net.RCNN_base[0].weight.grad is None
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
        cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 128


def test_fused_wgrad_sgd_equals_separate_update(cfg):
    """vrd.fc6-style skinny GEMM: SGD fused into the wgrad epilogue == wgrad followed by the SGD kernel (bitwise)."""
    from i2vsgg_amd import ops
    rng = np.random.default_rng(17)
    x = torch.from_numpy(rng.standard_normal((64, 9216), dtype=np.float32)).to(DEV)
    w0 = (rng.standard_normal((4096, 9216), dtype=np.float32) / 96).astype(np.float32)
    gy = torch.from_numpy(rng.standard_normal((64, 4096), dtype=np.float32)).to(DEV)
    res = []
    for fused in (False, True):
        w = torch.from_numpy(w0.copy()).to(DEV).requires_grad_()
        m = torch.from_numpy(rng.standard_normal(w0.shape, dtype=np.float32) * 0 + 0.01).to(DEV)
        if fused:
            ops.FUSED_SGD[w.data_ptr()] = (m, 1e-2, 0.9, 5e-4)
        try:
            y = ops.linear(x, w)
            y.backward(gy)
        finally:
            ops.FUSED_SGD.clear()
        if fused:
            assert w.grad is None
        else:
            ops.sgd_momentum_(w.data, w.grad, m, 1e-2, 0.9, 5e-4)
        res.append((w.detach().cpu().numpy().copy(), m.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert not np.array_equal(res[0][0], w0)


def test_sgg_step_schedules_match_single_graph(cfg):
    """Every schedule of the step computes the losses and weights of the eager sequential step: the sequential graph
    (overlap=False) and the overlapped graph (fork / join: head of batch k beside the backbone of batch k+1), each with
    the fused wgrad+SGD update and with the separate update kernels."""
    from i2vsgg_amd import train
    res = {}
    for key in (("eager", True), ("seq", True), ("stage", True), ("frame", True), ("none", True), ("seq", False), ("stage", False)):
        mode, fuse = key
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        overlapped = mode in ("stage", "frame", "none")      # how the backbone beside the head is cut into graph branches
        step = train.SGGEmbStep(net, 2, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5, fuse_sgd=fuse,
                                use_graph=mode != "eager", overlap=overlapped, bb_split=mode if overlapped else None)
        assert step.capture(warmup=1) == (mode != "eager"), step.graph_error
        assert step.overlap == overlapped and step.lag == {"stage": 2, "frame": 1, "none": 1}.get(mode, 0)
        losses = [float(step().item()) for _ in range(4)]
        step.opt.flush_pending()            # fc6 / fc7 hold their last update until the next forward (fused update only)
        torch.cuda.synchronize()
        res[key] = (losses, net.vrd.fc7.fc.weight.detach().cpu().numpy().copy())
        step.opt.unfuse()
    l0, w0 = res[("eager", True)]
    assert l0[0] != l0[3]                                                     # the weights do move
    for key, (l1, w1) in res.items():
        for a, b in zip(l0, l1):
            assert abs(a - b) <= 1e-5 * abs(a), (key, l0, l1)        # fp32 summation orders differ between the schedules
        _weights_close(w1, w0, "schedules %r" % (key,))


def test_sgg_step_is_bit_reproducible(cfg):
    """Round 5 (review item 4): no reduction of the relation step depends on arrival order any more.  Two runs of the step from
    equal weights on the same batch -- the captured / overlapped form bench.py times, and the eager form -- give the SAME BITS:
    five losses and every ``vrd.*`` tensor.  What it took: the split-K finish of the GEMM kernels sums any number of partials in
    split order (the relation head's skinny layers took fp32 atomics before), split filter gradients and the bias column sums meet
    in the caller's split workspace and are added in block order, the one launch left on the first-generation filter-gradient
    kernel (a 62-row layer's data gradient) is not split.  Full configs[1] shapes."""
    from i2vsgg_amd import train

    def run(graph):
        net = train.build_sgg_net(layers=101, seed=5, device=DEV)
        net.vrd.dropout = False
        step = train.SGGEmbStep(net, 2, seed=3, device=DEV, h=600, w=1000, n_boxes=32, n_pairs=32, use_graph=graph)
        if graph:
            assert step.capture(warmup=1, restore=True) and step.overlap, step.graph_error
        losses = torch.stack([step().clone() for _ in range(5)])
        step.opt.flush_pending()
        torch.cuda.synchronize()
        w = {k: v.detach().clone() for k, v in net.named_parameters() if k.startswith("vrd.")}
        step.opt.unfuse()
        return losses, w

    for graph in (True, False):
        la, wa = run(graph)
        lb, wb = run(graph)
        assert torch.equal(la, lb), (graph, (la - lb).tolist())
        assert float(la[0]) != float(la[4])                      # the weights do move
        diff = [k for k in wa if not torch.equal(wa[k], wb[k])]
        assert not diff, (graph, diff)


def test_instance_styled_step_is_bit_reproducible(cfg):
    """Round 6 (review item 3): the detector step too.  Its launch contexts are ``ordered`` now: split filter gradients of any
    part count (in-kernel finish up to 16 parts, side-by-side partial filters + a reduce pass beyond that and for the RPN's 18-row
    ``cls_score`` filter on the first-generation kernel), the Winograd-domain filter gradients (the final transform adds the
    parts), many-way split-K GEMMs and the bias column sums of netD_style's 37500-row projections (two levels of 32 row blocks)
    are all summed in a fixed order.  Two runs of the captured two-branch step and of the eager step from equal weights on the
    same minibatch: the same bits in every loss and every trainable tensor, and no reduction fell back to atomics."""
    from i2vsgg_amd import train
    from i2vsgg_amd._lib import lib

    def run(graph):
        torch.manual_seed(0)
        np.random.seed(3)
        net = train.build_instance_styled_net(50, device=DEV)
        step = train.InstanceStyleDStep(net, 1, seed=3, device=DEV, h=320, w=480)
        lib.i2v_ordered_fallbacks(1)
        if graph:
            assert step.capture(warmup=1, restore=True), step.graph_error
        losses = []
        for _ in range(4):
            step()
            losses.append(step._loss_buf.clone())
        torch.cuda.synchronize()
        assert lib.i2v_ordered_fallbacks(1) == 0, "a reduction that was asked to be ordered ran on atomics"
        w = {k: v.detach().clone() for k, v in net.named_parameters() if v.requires_grad}
        step.opt.unfuse()
        return torch.stack(losses), w

    for graph in (True, False):
        la, wa = run(graph)
        lb, wb = run(graph)
        assert torch.equal(la, lb), (graph, (la - lb).abs().max(dim=0).values.tolist())
        assert float(la[0, 0]) != float(la[3, 0])                # the weights do move
        diff = [k for k in wa if not torch.equal(wa[k], wb[k])]
        assert not diff, (graph, diff[:8], len(diff))


def test_sgg_step_back_to_back_replays_are_ordered(cfg, monkeypatch):
    """The overlapped step replayed back to back WITHOUT a host synchronisation between steps (the bench loop: the host
    runs several steps ahead of the device) follows the trajectory of the same step synchronised after every replay -- from
    an ordinary stream and from HIP's legacy default stream, with nothing but the graph launch on the caller's stream.
    Full configs[1] shapes: the ordering failure this guards against (the runtime's packet-capture
    replay path on the default stream) only shows when a step is long enough for the host to run ahead.

    The comparison is per step and per tensor: the step records, inside its own graph, checksums of what every head
    forward read and produced (``SGGEmbStep.trace``: loss, feature maps, scores, embedding, an RNG canary drawn from the
    dropout generator, fc7 / fc6 weights, boxes, labels).  A deviation is reported as (first step, first column) -- round 2
    saw ONE final loss off by 5.9e-5 with an abs-sum of the weights that could not say where it came from."""
    from i2vsgg_amd import ops, train
    n = 20
    # The default-stream runs below go through train.replay_graph's redirect (a private stream between two event edges),
    # whatever DEBUG_CLR_GRAPH_PACKET_CAPTURE says: the variable is no proof of what the runtime read (round-3 advice), so the
    # step does not consult it -- removed here to show that.  (Round 3 covered the redirect by starting tools/default_stream_probe.py
    # as a child of this process; a GPU-initialised process must not start programs on this pool.  The probe stays a top-level
    # program: profiles/r04_default_stream_probe.txt.)
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    train._REPLAY_STREAMS.clear()
    cols = train.SGGEmbStep.TRACE_COLS
    # relative tolerance per column: inputs that do not change are bit-stable, the RNG canary is exact, everything downstream
    # of the fp32 atomics of the small split-K GEMMs moves in the 7th digit (measured: loss 2.4e-7)
    tol = dict(loss=1e-5, features=1e-9, scores=1e-5, embedding=1e-5, rng_canary=0.0, fc7_weight=1e-7, fc6_weight_head=1e-7,
               boxes=0.0, labels=0.0)

    def run(own_stream, sync):
        net = train.build_sgg_net(101, device=DEV)
        step = train.SGGEmbStep(net, 2, seed=1, device=DEV, trace_rows=n + 8)
        prev = torch.cuda.current_stream()
        if own_stream:
            s = torch.cuda.Stream()
            s.wait_stream(prev)
            torch.cuda.set_stream(s)
        try:
            assert step.capture(warmup=2) and step.overlap, step.graph_error
            for i in range(3):
                step()
            torch.cuda.synchronize()            # bench.py's shape: warm-up, synchronise, timed steps back to back
            for i in range(n):
                step()
                if sync:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            assert int(step._trace_i) == n + 5                            # 2 warm-up + 3 + n head passes, one row each
            return step.trace[:n + 5].cpu().numpy(), float(step.loss), net.vrd.fc7.fc.weight.detach().clone()
        finally:
            torch.cuda.set_stream(prev)
            step.opt.unfuse()

    def first_deviation(a, b):
        for r in range(a.shape[0]):
            for c, name in enumerate(cols):
                if abs(a[r, c] - b[r, c]) > tol[name] * abs(a[r, c]):
                    return "step %d, %s: %.10g vs %.10g (rel %.2e)" % (r, name, b[r, c], a[r, c],
                                                                       abs(a[r, c] - b[r, c]) / max(abs(a[r, c]), 1e-300))
        return None

    want_t, want_l, want_w = run(True, True)
    assert want_t[2, 0] != want_t[-1, 0]                                 # the loss moves: the head does train
    for own in (True, False):
        for _rep in range(2):
            got_t, got_l, got_w = run(own, False)
            dev = first_deviation(want_t, got_t)
            assert dev is None, ("created stream" if own else "default stream", dev)
            assert abs(got_l - want_l) <= 1e-5 * abs(want_l), (own, got_l, want_l)
            _weights_close(got_w.cpu().numpy(), want_w.cpu().numpy(), "back to back, fc7, %s" % ("created" if own else "default"))
        # created stream: replayed where the caller stands; default stream: on the registry's "replay" stream
        assert bool(train._REPLAY_STREAMS) == (not own)
    h = train._REPLAY_STREAMS[torch.device(DEV).index].cuda_stream
    assert h != torch.cuda.default_stream(torch.device(DEV)).cuda_stream
    assert h == ops.stream_table()[(torch.device(DEV).index, "replay", 0)]


def test_sgg_step_staged_batches_meet_their_features(cfg):
    """A NEW minibatch staged between replays of the captured, overlapped step (what a training loop does; round-1
    advice: reseed() used to race the backbone pass already queued on the side stream): the loss sequence equals the
    eager sequential loop over the same batches -- batch k's boxes / labels always meet batch k's feature map -- with the
    lag of the pipeline (a batch staged before call k is the head's batch in call k+1 with the backbone cut by frame or not at
    all, in call k+2 with the backbone cut by stage: ``train.run_staged`` is the loop for any of them)."""
    from i2vsgg_amd import train
    seeds = [3, 11, 12, 13, 14]

    def run_eager():
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        step = train.SGGEmbStep(net, 2, seed=seeds[0], device=DEV, h=200, w=320, n_boxes=6, n_pairs=5, use_graph=False)
        assert not step.capture(warmup=1)
        losses = []
        for k, sd in enumerate(seeds):
            if k:
                step.reseed(sd)
            losses.append(float(step()))
        w = net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()
        step.opt.unfuse()
        return losses, w

    # the graph run reads each loss right after its call (the device scalar is overwritten by the next replay)
    def run_graph(split):
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        step = train.SGGEmbStep(net, 2, seed=seeds[0], device=DEV, h=200, w=320, n_boxes=6, n_pairs=5, bb_split=split)
        assert step.capture(warmup=1) and step.overlap, step.graph_error
        assert step.lag == (2 if split == "stage" else 1) and not step.bubble
        keep = torch.zeros(len(seeds), device=DEV)
        train.run_staged(step, [lambda sd=sd: step.reseed(sd) for sd in seeds[1:]], keep)
        # cut by stage: the second call of the run finds the first minibatch in the head's slot again and runs no head
        assert step.n_bubbles == (1 if split == "stage" else 0)
        torch.cuda.synchronize()
        w = net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()
        # a resident minibatch behind the run: every call trains it, none is a bubble
        before = float(step.loss)
        for _ in range(3):
            assert not step.bubble
            step()
        assert step.n_bubbles == (1 if split == "stage" else 0) and float(step.loss) != before
        step.opt.unfuse()
        return keep.tolist(), w

    l0, w0 = run_eager()
    assert len(set(round(x, 5) for x in l0)) == len(l0)                     # the batches differ
    for split in ("stage", "frame", "none"):
        l1, w1 = run_graph(split)
        for a, b in zip(l0, l1):
            assert abs(a - b) <= 1e-5 * abs(a), (split, l0, l1)   # a mismatched batch / feature pairing is off by 1e-1
        # eager: one backbone pass over both frames; graph: other launch shapes -- other split-K factors, other fp32 rounding
        _weights_close(w1, w0, "staged batches, fc7, captured (%s) vs eager" % split)


def test_captured_step_is_idempotent_after_one_warmup(cfg):
    """A graph captured right after ONE warm-up step (the arena of atomically accumulated outputs has just been
    re-sized and has recorded no use yet) still carries its clear: the backbone branch of every replay writes the eager
    feature map.  (Without the clear each replay accumulates into the previous one's outputs.)"""
    from i2vsgg_amd import train
    net = train.build_sgg_net(layers=50, seed=5, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5)
    assert step.capture(warmup=1) and step.overlap, step.graph_error
    torch.cuda.synchronize()
    with torch.no_grad():
        want = net.RCNN_base(step.im).clone()
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        assert _rel_err(step.fmap.cpu().numpy(), want.cpu().numpy()) < 1e-5
    step.opt.unfuse()


def test_winograd_filter_cache_follows_the_fused_optimizer(cfg):
    """Round-1 advice: FusedSGD writes parameters through raw device pointers (no Tensor._version bump), so a
    Winograd-domain filter cached by an eval forward must be invalidated by the optimizer: eval -> train steps -> eval
    equals the direct (cache-free) kernel on the updated weights."""
    from i2vsgg_amd import ops, train
    from i2vsgg_amd.model.faster_rcnn.layers import Bottleneck
    torch.manual_seed(0)
    blk = Bottleneck(256, 64).to(DEV)
    x = torch.randn(2, 256, 24, 40, device=DEV).contiguous(memory_format=torch.channels_last)
    opt = train.FusedSGD([("conv2.weight", blk.conv2.weight)], lr=0.5, momentum=0.0, weight_decay=0.0)
    with torch.no_grad():
        y0 = blk(x).clone()                               # caches the transformed filter
    for _ in range(2):
        opt.zero_grad()
        blk(x).square().mean().backward()
        opt.step()
    with torch.no_grad():
        y1 = blk(x)                                       # Winograd with the (re-)transformed filter
        s2, b2 = blk.bn2.folded()
        h = ops.conv2d(x, blk.conv1.weight, *blk.bn1.folded(), None, 1, 0, relu=True)
        h = ops.conv2d(h, blk.conv2.weight, s2, b2, None, 1, 1, relu=True)           # direct kernel, no cache
        want = ops.conv2d(h, blk.conv3.weight, *blk.bn3.folded(), x, 1, 0, relu=True)
    assert _rel_err(y1.cpu().numpy(), y0.cpu().numpy()) > 1e-3           # the weights did move
    assert _rel_err(y1.cpu().numpy(), want.cpu().numpy()) < 1e-4


def test_detect_frame_eval_loop_matches_oracle_postprocess(cfg):
    """eval.detect_frame (test_net_instance_styleD_bilinear.py:140-221): eval forward (TEST proposal settings) + the
    device post-processing pass == the oracle's restatement applied to the same network outputs."""
    from i2vsgg_amd import eval as ev
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    from oracle import rpn as orpn
    torch.manual_seed(1)
    net = resnet(tuple(range(16)), 50)
    net.create_architecture()
    net.to(DEV).eval()
    im, info = syn.frames(11, 1, 320, 480)
    imd, infod = torch.from_numpy(im).to(DEV), torch.from_numpy(info).to(DEV)
    z = torch.zeros(1, 1, 5, device=DEV)
    with torch.no_grad():
        out = net(imd, infod, z, torch.zeros(1, device=DEV))
    rois, cls_prob, bbox_pred = (t.cpu().numpy() for t in out[:3])
    assert rois.shape[1] == cfg.TEST.RPN_POST_NMS_TOP_N
    want = orpn.detection_postprocess(rois[0], cls_prob[0], bbox_pred[0], info[0, 0], info[0, 1], info[0, 2], False,
                                      cfg.TRAIN.BBOX_NORMALIZE_STDS, cfg.TRAIN.BBOX_NORMALIZE_MEANS, 0.0, cfg.TEST.NMS, 100)
    # the same network outputs for both sides: two forwards differ in the last bits (fp32 atomics in the small
    # split-K GEMMs of the head), which is enough to reorder near-tied scores
    got = ev.detect_frame(lambda *a: out, imd, infod, z, torch.zeros(1, device=DEV), thresh=0.0, max_per_image=100)
    live = ev.detect_frame(net, imd, infod, z, torch.zeros(1, device=DEV), thresh=0.0, max_per_image=100)
    assert len(live) == 16
    assert len(got) == 16 and sum(len(g) for g in got) > 0
    for j in range(16):
        assert np.array_equal(got[j], want[j]), j


@pytest.mark.parametrize("ic,gc", [(False, False), (True, True)])
def test_forward_detect_equals_the_eval_forward(cfg, ic, gc):
    """``forward_detect`` (what the test loop reads: rois, cls_prob, bbox_pred) == the same three outputs of ``forward`` in eval
    mode, with and without the context vectors of the two discriminators in the classifier input (split-K off: deterministic)."""
    from i2vsgg_amd._lib import TUNE, lib
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    torch.manual_seed(3)
    net = resnet(tuple(range(16)), 50, ic=ic, gc=gc)
    net.create_architecture()
    net.to(DEV).eval()
    im = torch.from_numpy(syn.frames(77, 1, 320, 480)[0]).to(DEV)
    info = torch.tensor([[320.0, 480.0, 1.3]], device=DEV)
    old = lib.i2v_get_tuning(TUNE["I2V_SPLIT_BELOW"])
    try:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], 0)
        with torch.no_grad():
            full = net(im, info, torch.zeros(1, 1, 5, device=DEV), torch.zeros(1, device=DEV))
        fast = net.forward_detect(im, info)
    finally:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], old)
    assert fast[1].shape == (1, cfg.TEST.RPN_POST_NMS_TOP_N, 16)
    for a, b in zip(fast, full[:3]):
        if ic or gc:      # the context vectors are spatial sums accumulated with fp32 atomics: two calls differ in the last bits
            assert _rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
        else:
            assert torch.equal(a, b)
    net.train()
    with pytest.raises(RuntimeError):
        net.forward_detect(im, info)


def test_detect_step_equals_frame_by_frame_eval(cfg):
    """eval.DetectStep (several frames per replayed graph, a branch per frame, im_info read on the device) returns what
    eval.detect_frame returns frame by frame: same boxes, same order, for frames of different scales, a short last batch and the
    pipelined ``run`` form.  Split-K is switched off for the comparison (its fp32 atomics reorder near-tied scores between any
    two forwards of the same frame)."""
    from i2vsgg_amd import eval as ev
    from i2vsgg_amd._lib import TUNE, lib
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    torch.manual_seed(1)
    net = resnet(tuple(range(16)), 50)
    net.create_architecture()
    net.to(DEV).eval()
    ims = [torch.from_numpy(syn.frames(40 + i, 1, 320, 480)[0]).to(DEV) for i in range(3)]
    infos = [torch.tensor([[320.0, 480.0, sc]], device=DEV) for sc in (1.0, 1.6, 0.8)]
    z, nb = torch.zeros(1, 1, 5, device=DEV), torch.zeros(1, device=DEV)
    old = lib.i2v_get_tuning(TUNE["I2V_SPLIT_BELOW"])
    try:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], 0)
        want = [ev.detect_frame(net, im, info, z, nb, thresh=0.0, max_per_image=100) for im, info in zip(ims, infos)]
        again = ev.detect_frame(net, ims[0], infos[0], z, nb, thresh=0.0, max_per_image=100)
        step = ev.DetectStep(net, frames=2, device=DEV)
        got = step(torch.cat(ims[:2]), torch.cat(infos[:2])) + step(ims[2], infos[2])          # the second call: one frame of two
        assert step.graph_error is None and step.shapes[step._staged].graph
        piped = [r for res in step.run([(torch.cat(ims[:2]), torch.cat(infos[:2])), (torch.cat(ims[1:]), torch.cat(infos[1:]))])
                 for r in res]
    finally:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], old)
    assert all(np.array_equal(a, b) for a, b in zip(again, want[0]))         # the comparison below is between deterministic runs
    assert len(got) == 3 and len(piped) == 4
    assert 0 < sum(len(c) for c in want[0]) <= 100
    for f, res in list(enumerate(got)) + [(0, piped[0]), (1, piped[1]), (1, piped[2]), (2, piped[3])]:
        assert len(res) == 16
        for j in range(16):
            assert np.array_equal(res[j], want[f][j]), (f, j)
    # the scale of a frame's im_info row reaches the boxes (:171 pred_boxes /= im_scale)
    assert max(c[:, :4].max() for c in want[2] if len(c)) > 1.5 * max(c[:, :4].max() for c in want[1] if len(c))


def test_relation_eval_branch_and_topk_vs_oracle(cfg):
    """eval branch of forward_relation (faster_rcnn_SGG_emb.py:583-697: all ordered pairs, union
    boxes, dual masks, relation head with softmax) and detection_output (lib/utils.py:584-628: confidence scaling +
    top-100 triplets) against the oracle's restatements."""
    from i2vsgg_amd import eval as ev
    from i2vsgg_amd.model.faster_rcnn.faster_rcnn_SGG_emb import build_eval_pair_tables
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import resnet
    from oracle import nets
    n_rel, n_cls = 62, 16
    torch.manual_seed(0)
    prd = syn.word_vectors(21, n_rel)
    net = resnet(tuple(range(n_cls)), _vrd_args(), 50, obj_vecs=syn.word_vectors(22, n_cls), prd_vecs=prd)
    net.create_architecture()
    params = syn.vrd_params(13)
    _load(net.vrd, params, "vrd.")
    net.to(DEV).eval()
    anno = syn.relation_annotation(31, 7, 8, n_rel, n_cls)
    net.vrd.target_gt_rels = {"t0": anno}
    ih, iw, sc = 600.0, 1000.0, 1.25
    # (1) pair tables == the reference's double loop over _getUnionBBox / _getDualMask
    boxes = np.array(anno["boxes"], np.float64) * sc
    union, bnd, ixs, ixo = build_eval_pair_tables(boxes, ih, iw)
    o_ixs, o_ixo, o_rel, o_masks = nets.eval_pair_tables(boxes, ih, iw, nets.union_box, nets.dual_mask)
    assert np.array_equal(ixs, o_ixs) and np.array_equal(ixo, o_ixo) and ixs.size == 7 * 6
    assert np.array_equal(union, o_rel[:, 1:])
    from i2vsgg_amd.model.faster_rcnn.faster_rcnn_SGG_emb import rasterize_masks
    assert np.array_equal(rasterize_masks(bnd, DEV).cpu().numpy(), o_masks.astype(np.float32))
    # (2) relation scores of the eval branch == oracle head in eval mode (softmax over predicates)
    fmap = np.abs(np.random.default_rng(32).standard_normal((1, 1024, 38, 63), dtype=np.float32))
    fm = torch.from_numpy(fmap).to(DEV).contiguous(memory_format=torch.channels_last)
    info = torch.tensor([[ih, iw, sc]], device=DEV)
    vrd_data = net.forward_relation_eval(fm, info, "t0")
    b5 = np.zeros((boxes.shape[0], 5), np.float32)
    b5[:, 1:] = boxes
    want, _ = nets.vrd_head(fmap, b5, o_rel.astype(np.float32), o_masks, o_ixs, o_ixo, prd,
                            {k: torch.as_tensor(v) for k, v in params.items()}, training=False)
    got = vrd_data["rel_score"].cpu().numpy()
    assert got.shape == (42, n_rel) and abs(got.sum(1) - 1).max() < 1e-5
    assert _rel_err(got, want.numpy()) < REL
    assert vrd_data["scores"] == [1] * 7 and vrd_data["bboxes"] == anno["boxes"]
    # (3) detection_output on the SAME scores: identical triplets, in order
    host = dict(vrd_data, rel_score=got)
    o = nets.detection_output(host, 100)
    g = ev.detection_output(vrd_data, 100)
    for a, b in zip(g, o):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # non-trivial confidences
    vrd_data["scores"] = list(np.random.default_rng(5).uniform(0.3, 1.0, 7).astype(np.float32))
    host = dict(vrd_data, rel_score=got)
    for a, b in zip(ev.detection_output(vrd_data, 100), nets.detection_output(host, 100)):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # degenerate frames (:588-594)
    net.vrd.target_gt_rels = {"one": {"boxes": [[1, 2, 30, 40]], "box_classes": [3], "rels": []},
                              "none": {"boxes": [], "box_classes": [], "rels": []}}
    assert ev.detection_output(net.forward_relation_eval(fm, info, "one")) == (None,) * 5
    assert net.forward_relation_eval(fm, info, "none") == {"bboxes": [], "classes": [], "scores": []}


def test_relation_step_equals_frame_by_frame_eval(cfg):
    """eval.RelationStep (several frames per replayed graph, boxes / pairs padded to a capacity) returns what
    eval.relation_frame + detection_output return frame by frame: same triplets, same order, same confidences -- for frames
    with different numbers of boxes, a degenerate frame, a capacity growth and the pipelined ``run`` form.  Split-K off, as in
    the detector's test."""
    from i2vsgg_amd import eval as ev
    from i2vsgg_amd._lib import TUNE, lib
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import resnet
    n_rel, n_cls = 62, 16
    torch.manual_seed(0)
    net = resnet(tuple(range(n_cls)), _vrd_args(), 50, obj_vecs=syn.word_vectors(22, n_cls), prd_vecs=syn.word_vectors(21, n_rel))
    net.create_architecture()
    _load(net.vrd, syn.vrd_params(13), "vrd.")
    net.to(DEV).eval()
    H, W = 320, 480
    ims = [torch.from_numpy(syn.frames(50 + i, 1, H, W)[0]).to(DEV) for i in range(4)]
    infos = np.array([[H, W, 1.0], [H, W, 1.25], [H, W, 0.8], [H, W, 1.0]], np.float32)
    nbox = [5, 3, 1, 9]
    net.vrd.target_gt_rels = {"f%d" % i: syn.relation_annotation(60 + i, nbox[i], min(nbox[i], nbox[i] * (nbox[i] - 1)), n_rel, n_cls, im_h=int(H / infos[i, 2]),
                                                                 im_w=int(W / infos[i, 2])) for i in range(4)}
    old = lib.i2v_get_tuning(TUNE["I2V_SPLIT_BELOW"])
    try:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], 0)
        want = [ev.relation_frame(net, ims[i], torch.from_numpy(infos[i:i + 1]).to(DEV), "f%d" % i)[1] for i in range(4)]
        step = ev.RelationStep(net, frames=2, device=DEV, cap_boxes=6)
        got = step(torch.cat(ims[:2]), infos[:2], ["f0", "f1"])
        assert step.graph_error is None and step.shapes[step._staged].graph and step.cap_boxes == 6
        got += step(torch.cat(ims[2:]), infos[2:], ["f2", "f3"])              # 9 boxes: the capacity grows, the graph is captured again
        assert step.cap_boxes == 10 and step.shapes[step._staged].graph
        piped = [r for res in step.run([(torch.cat(ims[:2]), infos[:2], ["f0", "f1"]), (ims[3], infos[3:], ["f3"])]) for r in res]
    finally:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], old)
    assert want[2] == (None,) * 5 and got[2] == (None,) * 5
    for f, res in list(enumerate(got)) + [(0, piped[0]), (1, piped[1]), (3, piped[2])]:
        if f == 2:
            continue
        for a, b in zip(res, want[f]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), f
    assert len(want[1][1]) == 100 and want[1][0][:, 1].max() > 0


def test_extract_feature_and_box_classification_vs_oracle(cfg):
    """``_extract_feature`` (faster_rcnn_SGG_emb.py:381-392) and the box classification of the eval branch (:278-291):
    roi_layers.ROIAlign -> layer4 -> mean -> RCNN_cls_score -> softmax with the background column zeroed."""
    from i2vsgg_amd.model.faster_rcnn.resnet_SGG_emb import resnet
    from oracle import cops, nets
    n_cls = 16
    torch.manual_seed(0)
    net = resnet(tuple(range(n_cls)), _vrd_args(), 101, obj_vecs=syn.word_vectors(22, n_cls), prd_vecs=syn.word_vectors(21, 62))
    net.create_architecture()
    p = dict(syn.backbone_params(0, 101), **syn.det_head_params(11, n_cls))
    _load(net.RCNN_top, p, "RCNN_top.")
    _load(net.RCNN_cls_score, p, "RCNN_cls_score.")
    net.to(DEV).eval()
    fmap = np.abs(np.random.default_rng(5).standard_normal((1, 1024, 38, 63), dtype=np.float32))
    boxes = syn.boxes(9, 6, 600, 1000)
    fm = torch.from_numpy(fmap).to(DEV).contiguous(memory_format=torch.channels_last)
    got = net._extract_feature(fm, boxes)
    r5 = np.hstack((np.zeros((6, 1), np.float32), boxes.astype(np.float32)))
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    pooled = torch.from_numpy(cops.roi_align_sampled_fwd(fmap, r5, 7, 7, 1.0 / 16.0, 0))
    want = nets.head_to_tail(pooled, tp)
    assert got.shape == (6, 2048) and _rel_err(got, want.numpy()) < REL
    assert _rel_err(net._extract_feature(fmap, boxes), want.numpy()) < REL           # numpy map in, as the reference passes it
    prob = torch.softmax(want @ tp["RCNN_cls_score.weight"].t() + tp["RCNN_cls_score.bias"], 1)
    prob[:, 0] = 0
    cls, conf = net.classify_boxes(fm, boxes)
    assert np.array_equal(cls, prob.argmax(1).numpy()) and np.allclose(conf, prob.max(1).values.numpy(), atol=1e-5)


def test_consistency_terms_match_reference_formula(cfg):
    """--cr (trainval_net_instance_styleD_bilinear.py:299-311) with the reference's literal expressions at 128 ROIs."""
    from i2vsgg_amd import train
    g = torch.Generator().manual_seed(3)
    B, R = 2, 128
    d_inst, d_inst_t = torch.rand(B * R, 1, 7, 7, generator=g), torch.rand(B * R, 1, 7, 7, generator=g)
    d_style, d_style_t = torch.rand(B, 1, generator=g), torch.rand(B, 1, generator=g)
    mse = torch.nn.MSELoss()
    want_s = mse(torch.mean(torch.mean(d_inst, dim=3), dim=2), d_style.repeat(1, 128).view(-1, 1).detach())
    want_t = mse(torch.mean(torch.mean(d_inst_t, dim=3), dim=2), d_style_t.repeat(1, 128).view(-1, 1).detach())
    got = train.consistency_terms(d_inst, d_style, d_inst_t, d_style_t)
    assert torch.equal(got["source_adv_cst"], want_s) and torch.equal(got["target_adv_cst"], want_t)


def test_sgg_step_tensor_parallel_fc6_rehearsal_matches_single_graph(cfg, monkeypatch):
    """The multi-GPU form of the step on ONE rank (1-rank RCCL group, I2V_FORCE_EXCHANGE=1): column-parallel fc6 with
    its three collectives and the all-reduce of the remaining gradients captured in the head branch of the step graph, fused
    fc6 update -- same losses and weights as the single-GPU step.  (The 2-rank numerics of the column cut are checked on CPU under gloo.)"""
    import torch.distributed as dist
    from i2vsgg_amd import parallel, train
    res = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("I2V_FORCE_EXCHANGE", "1")
            monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
            monkeypatch.setenv("MASTER_PORT", str(29600 + os.getpid() % 300))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        try:
            net = train.build_sgg_net(layers=50, seed=5, device=DEV)
            net.vrd.dropout = False
            step = train.SGGEmbStep(net, 1, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5)
            assert step.tp == forced and (net.vrd.tp is not None) == forced
            assert "vrd.fc6.fc.weight" in step.fused           # the fc6 update stays fused in both forms
            assert step.capture(warmup=1) and step.overlap, step.graph_error
            losses = [float(step().item()) for _ in range(3)]
            torch.cuda.synchronize()
            w6, _ = net.vrd.gather_fc6()
            res.append((losses, w6.detach().cpu().numpy().copy(), net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()))
            step.opt.unfuse()
        finally:
            # the graphs that captured RCCL kernels go before the communicator
            step = net = None
            import gc
            gc.collect()
            torch.cuda.synchronize()
            if forced:
                dist.destroy_process_group()
    (l0, a0, b0), (l1, a1, b1) = res
    assert l0[0] != l0[2]
    for x, y in zip(l0, l1):
        assert abs(x - y) <= 1e-5 * abs(x), (l0, l1)
    _weights_close(a1, a0, "fc6, column-parallel rehearsal vs single graph")
    _weights_close(b1, b0, "fc7, column-parallel rehearsal vs single graph")


@pytest.mark.parametrize("which", ["sgg_plain_dp", "instance_styled"])
def test_rccl_rehearsal_of_the_plain_data_parallel_exchanges(cfg, monkeypatch, which):
    """Round-4 review item 8: the two RCCL-in-graph paths that had never run on a GPU.  One rank, a 1-rank ``nccl`` group,
    I2V_FORCE_EXCHANGE=1:
      * ``sgg_plain_dp``: SGGEmbStep with I2V_TP_FC6=0 -- north_star's literal partitioning, ONE sum all-reduce of every
        ``vrd.*`` gradient (906 MB, the 822 MB fc6 gradient first) captured in the head branch of the step graph;
      * ``instance_styled``: InstanceStyleDStep captured with its 202 MB all-reduce (train.py ``_body_branches``: after the
        join of the source / target branches, before the update).
    Same losses and weights as the single-GPU captured step from the same seeds; the graphs are dropped before the
    communicator.  (World size 1: the reduction is the identity -- what is rehearsed is the capture of the collectives, the
    bucket packing and the schedule; the 2-rank numerics run on CPU under gloo.  Multi-GPU: unmeasured on hardware.)"""
    import gc
    import torch.distributed as dist
    from i2vsgg_amd import parallel, train
    res = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("I2V_FORCE_EXCHANGE", "1")
            monkeypatch.setenv("I2V_TP_FC6", "0")
            monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
            monkeypatch.setenv("MASTER_PORT", str(29900 + os.getpid() % 90))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        step = net = None
        try:
            torch.manual_seed(0)
            np.random.seed(3)
            assert parallel.exchange_enabled() == forced
            if which == "sgg_plain_dp":
                net = train.build_sgg_net(layers=50, seed=5, device=DEV)
                net.vrd.dropout = False
                step = train.SGGEmbStep(net, 1, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5)
                assert not step.tp and net.vrd.tp is None
                # an exchanged gradient cannot be consumed by a fused update: fc6 is fused only where its gradient stays local
                assert ("vrd.fc6.fc.weight" in step.fused) == (not forced)
                assert step.capture(warmup=1) and step.overlap, step.graph_error
                losses = [float(step().item()) for _ in range(3)]
                step.opt.flush_pending()
                watch = [net.vrd.fc6.fc.weight, net.vrd.fc7.fc.weight, net.vrd.fc_rel.fc.bias]
            else:
                net = train.build_instance_styled_net(50, device=DEV)
                step = train.InstanceStyleDStep(net, 1, seed=3, device=DEV, h=256, w=320)
                assert step.capture(warmup=1), step.graph_error
                losses = []
                for _ in range(3):
                    step()
                    losses.append(float(step.losses["total"]))
                p = dict(net.named_parameters())
                watch = [p["RCNN_base.6.5.conv3.weight"], p["RCNN_rpn.RPN_Conv.weight"], p["netD_style.fc_1.weight"],
                         p["RCNN_bbox_pred.bias"]]
            torch.cuda.synchronize()
            res.append((losses, [w.detach().cpu().numpy().copy() for w in watch]))
            step.opt.unfuse()
        finally:
            step = net = watch = None                # the graphs that captured RCCL kernels go before the communicator
            gc.collect()
            torch.cuda.synchronize()
            if forced:
                dist.destroy_process_group()
    (l0, w0), (l1, w1) = res
    assert l0[0] != l0[2] and all(np.isfinite(l0)) and all(np.isfinite(l1))
    for x, y in zip(l0, l1):
        assert abs(x - y) <= 1e-5 * abs(x), (l0, l1)
    # the relation step's sums are ordered (bit-reproducible); the instance_styleD step keeps round 4's rule (fp32 atomics in its
    # many-way split reductions: ordering them costs 4 % of its step), so two of ITS runs agree to rounding, not to the bit
    for i, (a, b) in enumerate(zip(w0, w1)):
        _weights_close(b, a, "%s rehearsal vs single graph, tensor %d" % (which, i), peak=1e-5)       # both steps' sums are ordered (round 6)


def test_fork_inside_a_graph_branch_is_an_error_not_a_crash():
    """Round-2 review: a fork made inside a forked branch ends hipStreamEndCapture in a host segfault on ROCm 7.2, and the
    step objects avoided it by construction only.  They fork through ``ops.branch`` now, which refuses the nested fork while a
    capture is running (and is an ordinary fork / join outside one)."""
    from i2vsgg_amd import ops
    s1, s2 = ops.role_stream(DEV, ("frame", 0)), ops.role_stream(DEV, ("frame", 1))
    x = torch.ones(1024, device=DEV)
    torch.cuda.synchronize()
    g, caught = torch.cuda.CUDAGraph(), []
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        with ops.branch(s1, main):
            y = x * 2
            try:
                with ops.branch(s2, s1):
                    y = y + 1
            except RuntimeError as e:
                caught.append(str(e))
        ops.join(main, s1)
        z = y + 1
    g.replay()
    torch.cuda.synchronize()
    assert caught and "fork inside a forked graph branch" in caught[0]
    assert float(z[0]) == 3.0
    main = torch.cuda.current_stream()              # outside a capture nesting is an ordinary (if pointless) fork / join
    with ops.branch(s1, main):
        with ops.branch(s2, s1):
            w = x + 5
        ops.join(s1, s2)
    ops.join(main, s1)
    torch.cuda.synchronize()
    assert float(w[0]) == 6.0


def test_branch_streams_never_alias(cfg):
    """Round-3 review: torch.cuda.Stream() deals 32 pooled handles round robin, so after a few step objects a "copy" or branch
    stream could BE the stream a later capture forks or captures on.  The step objects take their streams from ops.role_stream:
    one HIP stream per role, created once per process by the library, distinct from each other and from every pooled handle;
    ops.branch refuses an alias of the forking stream or of an open sibling."""
    from i2vsgg_amd import ops, train
    dev = torch.device(DEV)
    pooled = {torch.cuda.Stream(dev).cuda_stream for _ in range(40)}          # the whole pool, dealt round robin
    assert len(pooled) <= 32
    pooled |= {torch.cuda.default_stream(dev).cuda_stream, torch.cuda.current_stream(dev).cuda_stream}
    net = train.build_sgg_net(layers=50, seed=5, device=DEV)
    steps = [train.SGGEmbStep(net, 2, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5) for _ in range(3)]
    table = ops.stream_table()
    handles = list(table.values())
    assert len(set(handles)) == len(handles) and not (set(handles) & pooled), table
    for a in steps[1:]:                                  # objects share the registry's streams instead of drawing new ones
        assert [t.cuda_stream for t in a._frame_streams] == [t.cuda_stream for t in steps[0]._frame_streams]
    assert ops.role_stream(dev, ("frame", 0)) is ops.role_stream(dev, ("frame", 0))
    for st in steps:
        st.opt.unfuse()
    main = ops.role_stream(dev, "warmup")                # (ExternalStream(0) would draw a pooled stream, so not the default stream)
    with torch.cuda.stream(main):
        alias = torch.cuda.ExternalStream(main.cuda_stream, device=dev)
        assert alias.cuda_stream == main.cuda_stream
        with pytest.raises(RuntimeError, match="IS the forking stream"):
            with ops.branch(alias, main):
                pass
        s1 = ops.role_stream(dev, ("frame", 0))
        twin = torch.cuda.ExternalStream(s1.cuda_stream, device=dev)
        with ops.branch(s1, main):
            pass
        with pytest.raises(RuntimeError, match="already an open branch"):
            with ops.branch(twin, main):
                pass
        ops.join(main, s1)
        with ops.branch(twin, main):                     # joined: the handle is free again
            pass
        ops.join(main, twin)
    torch.cuda.synchronize()


def test_sgg_step_with_adam_captured_equals_eager(cfg):
    """``--o adam`` (trainval_net_SGG_emb.py:146-147): the relation step with train.FusedAdam, captured (overlapped graph) against
    eager launches -- the step count sits in device memory, so every replay applies its own bias corrections."""
    from i2vsgg_amd import train
    res = {}
    for mode in ("eager", "overlap"):
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        step = train.SGGEmbStep(net, 2, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5, use_graph=mode != "eager",
                                overlap=mode == "overlap", optimizer="adam")
        assert isinstance(step.opt, train.FusedAdam) and step.fused == []
        assert step.capture(warmup=1, restore=True) == (mode != "eager"), step.graph_error
        assert int(step.opt.t.item()) == 0                          # the warm-up step was undone, its step count included
        losses = [float(step().item()) for _ in range(4)]
        torch.cuda.synchronize()
        assert int(step.opt.t.item()) == 4                          # one optimizer step per call, in both schedules
        res[mode] = (losses, net.vrd.fc7.fc.weight.detach().cpu().numpy().copy())
    l0, w0 = res["eager"]
    l1, w1 = res["overlap"]
    assert l0[0] != l0[3]
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * abs(a), (l0, l1)
    assert np.isfinite(w1).all()
