"""instance_styleD D+G step: the HIP model against the CPU oracle composed end to end
(backbone -> netD_style -> RPN -> anchor targets -> proposal targets -> RoIAlignAvg -> netD_pixel ->
layer4 -> cls/bbox losses), same seeded weights, same np.random stream.  Losses within 1e-3 relative
(BASELINE.json north_star: "within 1e-3 rel fp32 on ... D/G losses")."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from i2vsgg_amd import synthetic as syn  # noqa: E402

DEV = "cuda:0"
REL = 1e-3
# Bounds of the loose comparisons, set from what one run records (profiles/r06_parity_margins.txt; the tests append what they see
# to gpurun_out/parity_margins.txt on every run): proposal sets of 1000-2000 rows overlap 0.9883-0.9967 -> 0.98 (a margin of
# 0.008: ~16 swapped near-ties of 2000); the 32-row target sets 31/32-32/32 -> at most TWO rows may differ; the four
# sampling-dependent losses of the model as shipped deviate <= 4.1e-6 (the sampled rows coincide) -> north_star's 1e-3, like
# the other four (rounds 1-5 allowed 1e-2 and 0.95 / 0.90 without knowing the actual values).
SET_OVERLAP = 0.98
AS_SHIPPED = 1e-3


def min_overlap(n):
    """Rows of an n-row reference proposal set the HIP model must reproduce."""
    return SET_OVERLAP * n if n >= 500 else n - 2


from conftest import record_margin  # noqa: E402


def test_instance_styled_source_and_target_losses_vs_oracle():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    from i2vsgg_amd.model.utils import config as c
    from oracle import cops, nets, rpn
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                     "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
    cfg = c.cfg
    try:
        B, H, W, n_cls = 2, 320, 480, 16
        p = {}
        p.update(syn.backbone_params(0, 50, top=True))
        p.update(syn.rpn_params(10, std=0.02))
        p.update(syn.det_head_params(11, n_cls))
        p.update(syn.netd_params(12))
        net = resnet(tuple(range(n_cls)), 50)
        net.create_architecture()
        r = load_reference_state(net, p, strict=False)
        assert not r.unexpected_keys and all("num_batches" in k for k in r.missing_keys), (r.missing_keys[:5], r.unexpected_keys[:5])
        net.to(DEV).train()
        im, info = syn.frames(5, B, H, W)
        gt, nb = syn.gt_boxes(6, B, 6, n_cls, im_h=H, im_w=W)
        imd, infod = torch.from_numpy(im).to(DEV), torch.from_numpy(info).to(DEV)
        gtd, nbd = torch.from_numpy(gt).to(DEV), torch.from_numpy(nb).to(DEV)

        stash = {}
        rpn_forward = net.RCNN_rpn.forward

        def spy(*a, **k):
            out = rpn_forward(*a, **k)
            stash["rois"] = out[0].detach().cpu().numpy()
            return out
        net.RCNN_rpn.forward = spy
        np.random.seed(3)
        rois, cls_prob, bbox_pred, l_rpn_cls, l_rpn_box, l_cls, l_box, labels, d_inst, d_sty = net(
            imd, infod, gtd, nbd, target=False, eta=0.1, eta_style=0.001)
        src_rois = stash["rois"]
        d_inst_t, d_sty_t = net(imd, infod, torch.zeros(B, 1, 5, device=DEV), torch.zeros(B, device=DEV), target=True,
                                eta=0.1, eta_style=0.001)
        tgt_rois = stash["rois"]

        # ---------------- oracle, same weights, same RNG stream
        po = {k: v.clone() for k, v in p.items()}
        with torch.no_grad():
            feat, feat1 = nets.extract_feature(torch.from_numpy(im), po, blocks=(3, 4, 6))
            d_sty_o = nets.netd_style(feat1, po, 0.001)
            cls, prob, box = nets.rpn_head(feat, po)
        fh, fw = feat.shape[2], feat.shape[3]
        rois_o, _ = rpn.proposal_layer(prob[:, 9:].numpy(), box.numpy(), info, 12000, 2000, 0.7)
        # proposal stage: same boxes up to score near-ties (conv rounding); compared as row sets
        for b in range(B):
            a = {tuple(np.round(x, 1)) for x in src_rois[b] if x[1:].any()}
            o = {tuple(np.round(x, 1)) for x in rois_o[b] if x[1:].any()}
            record_margin("source_and_target_losses_vs_oracle", "proposal set overlap, frame %d (of %d)" % (b, len(o)), len(a & o) / len(o), min_overlap(len(o)) / len(o))
            assert len(a & o) >= min_overlap(len(o)), (len(a & o), len(o))
        rs = np.random.RandomState(3)
        L, T, IW, OW = rpn.anchor_target_layer(fh, fw, gt, info, rs)
        pair = cls.view(B, 2, 9 * fh, fw).permute(0, 2, 3, 1).reshape(-1, 2)
        lab = torch.from_numpy(L).reshape(-1)
        keep = lab.ne(-1).nonzero().view(-1)
        o_rpn_cls = F.cross_entropy(pair[keep], lab[keep].long()).item()
        o_rpn_box = rpn.smooth_l1(box.numpy(), T, IW, OW, sigma=3, sum_dims=(1, 2, 3))
        # downstream of the proposals: feed the HIP path's own proposals so that the sampled set is identical
        rois_b, labels_o, tg, inw, outw = rpn.proposal_target_layer(src_rois, gt, rs, batch_size=32)
        assert np.array_equal(rois_b, rois.cpu().numpy()) and np.array_equal(labels_o.reshape(-1), labels.cpu().numpy())
        pooled = torch.from_numpy(cops.roi_align_avg_fwd(feat.numpy(), rois_b.reshape(-1, 5), 7, 7, 1.0 / 16.0))
        with torch.no_grad():
            d_inst_o = nets.netd_pixel(pooled, po, 0.1)
            h = nets.head_to_tail(pooled, po, nblocks=3)
            bp = F.linear(h, po["RCNN_bbox_pred.weight"], po["RCNN_bbox_pred.bias"])
            lo = torch.from_numpy(labels_o.reshape(-1)).long()
            bp = torch.gather(bp.view(-1, n_cls, 4), 1, lo.view(-1, 1, 1).expand(-1, 1, 4)).squeeze(1)
            cs = F.linear(h, po["RCNN_cls_score.weight"], po["RCNN_cls_score.bias"])
            o_cls = F.cross_entropy(cs, lo).item()
        o_box = rpn.smooth_l1(bp.numpy(), tg.reshape(-1, 4), inw.reshape(-1, 4), outw.reshape(-1, 4))
        pooled_t = torch.from_numpy(cops.roi_align_avg_fwd(feat.numpy(), tgt_rois.reshape(-1, 5), 7, 7, 1.0 / 16.0))
        with torch.no_grad():
            d_inst_to = nets.netd_pixel(pooled_t, po, 0.1)

        def rel(a, b):
            return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)

        # the eight scalars of trainval_net_instance_styleD_bilinear.py:276-296
        checks = {
            "rpn_loss_cls": (l_rpn_cls.mean().item(), o_rpn_cls),
            "rpn_loss_box": (l_rpn_box.mean().item(), o_rpn_box),
            "RCNN_loss_cls": (l_cls.mean().item(), o_cls),
            "RCNN_loss_bbox": (l_box.mean().item(), o_box),
            "dloss_s (instance)": (0.5 * torch.mean(d_inst ** 2).item(), 0.5 * torch.mean(d_inst_o ** 2).item()),
            "dloss_s_style": (0.5 * torch.mean(d_sty ** 2).item(), 0.5 * torch.mean(d_sty_o ** 2).item()),
            "dloss_t (instance)": (0.5 * torch.mean((1 - d_inst_t) ** 2).item(), 0.5 * torch.mean((1 - d_inst_to) ** 2).item()),
            "dloss_t_style": (0.5 * torch.mean((1 - d_sty_t) ** 2).item(), 0.5 * torch.mean((1 - d_sty_o) ** 2).item()),
        }
        for name, (got, ref) in checks.items():
            assert rel(got, ref) < REL, (name, got, ref)
        np.testing.assert_allclose(d_inst.detach().cpu().numpy(), d_inst_o.numpy(), rtol=REL, atol=1e-6)
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
        cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 128


@pytest.mark.parametrize("tag,B,H,W", [("small", 2, 320, 480), ("full", 1, 600, 1000)])
def test_instance_styled_step_vs_reference_run_golden(gold, tag, B, H, W):
    """the eight scalars of one D+G step (trainval_net_instance_styleD_bilinear.py:276-296)
    against the REFERENCE's own ``_fasterRCNN.forward`` run on the same seeded weights and frames
    (tests/golden/instance_styled_step.npz: its backbone, RPN + nms_cpu, target layers under np.random.seed(3),
    discriminators and heads; RoIAlignAvg = its own compiled ROIAlignForwardCpu + avg_pool2d).  res101; 2+2 frames of
    320x480 and 1+1 full 600x1000 frames, 32 ROIs per frame (configs[2]'s shapes).

    Two passes.  (1) The HIP model as shipped: its proposals against the reference's as row sets (a near-tie between two
    scores may order two boxes differently under another convolution rounding), the two RPN losses and the style terms --
    which do not depend on the proposals -- within 1e-3.  (2) With the reference's proposals handed to the sampling layer
    (the RPN still runs: its losses and its share of the np.random stream are the HIP path's own): sampled rois and labels
    bit-equal, all eight scalars, cls_prob, bbox_pred and the discriminator maps within 1e-3."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    from i2vsgg_amd.model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    from i2vsgg_amd.model.utils import config as c
    g = gold("instance_styled_step")
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                     "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
    cfg = c.cfg
    try:
        n_cls = 16
        net = resnet(tuple(range(n_cls)), 101)
        net.create_architecture()
        r = load_reference_state(net, syn.instance_styled_step_params(), strict=False)
        assert not r.unexpected_keys and all("num_batches" in k for k in r.missing_keys)
        net.to(DEV).train()
        im, info, gt, nb, im_t, info_t = syn.instance_styled_step_inputs(B, H, W, n_cls)
        dev = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
        rpn_forward = net.RCNN_rpn.forward
        seen, inject = [], {}

        def spy(*a, **k):
            out = rpn_forward(*a, **k)
            seen.append(out[0].detach().cpu().numpy())
            if inject:
                return (dev(inject.pop("rois")),) + tuple(out[1:])
            return out
        net.RCNN_rpn.forward = spy

        def step():
            np.random.seed(3)
            with torch.no_grad():
                o = net(dev(im), dev(info), dev(gt), dev(nb), target=False, eta=0.1, eta_style=0.001)
                t = net(dev(im_t), dev(info_t), torch.zeros(B, 1, 5, device=DEV), torch.zeros(B, device=DEV), target=True,
                        eta=0.1, eta_style=0.001)
            rois, cls_prob, bbox_pred, l_rpn_cls, l_rpn_box, l_cls, l_box, labels, d_inst, d_sty = o
            losses = np.array([float(v) for v in (
                l_rpn_cls.mean(), l_rpn_box.mean(), l_cls.mean(), l_box.mean(), 0.5 * torch.mean(d_inst ** 2),
                0.5 * torch.mean(d_sty ** 2), 0.5 * torch.mean((1 - t[0]) ** 2), 0.5 * torch.mean((1 - t[1]) ** 2))])
            return losses, o, t

        names = [str(n) for n in g["loss_names"]]
        want = g[tag + "_losses"]

        # (1) as shipped
        losses, _, _ = step()
        for b in range(B):
            for got, ref in ((seen[0][b], g[tag + "_rpn_rois_src"][b]), (seen[1][b], g[tag + "_rpn_rois_tgt"][b])):
                a = {tuple(np.round(x, 1)) for x in got if x[1:].any()}
                o = {tuple(np.round(x, 1)) for x in ref if x[1:].any()}
                record_margin("reference_run_golden[%s]" % tag, "proposal set overlap, frame %d (of %d)" % (b, len(o)), len(a & o) / len(o), min_overlap(len(o)) / len(o))
                assert len(a & o) >= min_overlap(len(o)), (len(a & o), len(o))
        for i in range(8):                                  # the four that depend on which rows were sampled: AS_SHIPPED (1e-3 too)
            tol = REL if i in (0, 1, 5, 7) else AS_SHIPPED
            record_margin("reference_run_golden[%s]" % tag, "as shipped: " + names[i], abs(losses[i] - want[i]) / abs(want[i]), tol)
            assert abs(losses[i] - want[i]) <= tol * abs(want[i]), (names[i], losses[i], want[i])
        as_shipped = losses

        # (2) the reference's proposals in front of the sampling layer
        del seen[:]
        inject["rois"] = g[tag + "_rpn_rois_src"]
        np.random.seed(3)
        with torch.no_grad():
            o = net(dev(im), dev(info), dev(gt), dev(nb), target=False, eta=0.1, eta_style=0.001)
            inject["rois"] = g[tag + "_rpn_rois_tgt"]
            t = net(dev(im_t), dev(info_t), torch.zeros(B, 1, 5, device=DEV), torch.zeros(B, device=DEV), target=True,
                    eta=0.1, eta_style=0.001)
        rois, cls_prob, bbox_pred, l_rpn_cls, l_rpn_box, l_cls, l_box, labels, d_inst, d_sty = o
        assert np.array_equal(rois.cpu().numpy(), g[tag + "_rois"])
        assert np.array_equal(labels.cpu().numpy(), g[tag + "_labels"])
        losses = np.array([float(v) for v in (
            l_rpn_cls.mean(), l_rpn_box.mean(), l_cls.mean(), l_box.mean(), 0.5 * torch.mean(d_inst ** 2),
            0.5 * torch.mean(d_sty ** 2), 0.5 * torch.mean((1 - t[0]) ** 2), 0.5 * torch.mean((1 - t[1]) ** 2))])
        for i, n in enumerate(names):
            assert abs(losses[i] - want[i]) <= REL * abs(want[i]), (n, losses[i], want[i])
        for got, key in ((cls_prob, "_cls_prob"), (bbox_pred, "_bbox_pred"), (d_inst, "_d_instance"), (d_sty, "_d_style"),
                         (t[0], "_d_instance_t"), (t[1], "_d_style_t")):
            ref = g[tag + key]
            np.testing.assert_allclose(got.cpu().numpy().reshape(ref.shape), ref, rtol=REL, atol=REL * np.abs(ref).max())
        print("\n%s: reference %s\n  HIP, reference proposals %s\n  HIP as shipped %s" % (
            tag, np.round(want, 6), np.round(losses, 6), np.round(as_shipped, 6)))
    finally:
        cfg.TRAIN.BATCH_SIZE = 128
        cfg.TRAIN.RPN_POST_NMS_TOP_N_TARGET = 128
