"""Parity of the HIP kernels (called through the C-ABI of libi2vsgg_hip.so) against the CPU
oracle and the reference-generated golden vectors.  Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from i2vsgg_amd import synthetic as syn  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (torch.cuda.is_available() is False)")
    from i2vsgg_amd import ops as o
    return o


@pytest.fixture(scope="module")
def oracle():
    from oracle import cops, rpn
    return cops, rpn


DEV = "cuda:0"


def _rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _rois_cases(rng, B, H, W, scale=16.0, n=24):
    """ROIs incl. degenerate (x2<x1), sub-pixel, out-of-image and full-image boxes."""
    return syn.roi_cases(int(rng.integers(1 << 30)), B, H, W, scale, n)


@pytest.mark.parametrize("C,H,W,B", syn.ROI_ALIGN_GOLDEN_CASES)
def test_roi_align_fwd_bit_equal_to_the_reference_c(ops, gold, C, H, W, B):
    """The HIP RoIAlign / RoIAlignAvg against outputs of the reference's OWN ``ROIAlignForwardCpu``
    (roi_align/src/roi_align.c:80-136; tests/golden/roi_align_fwd.npz, tier "extracted") and its
    ``avg_pool2d(2, 1)`` (modules/roi_align.py:27-29): same bytes for both map layouts and both output layouts, on every
    ROI class, incl. the full-size 1024-channel map (the ROI-per-workgroup kernel) and the narrow ones (the column kernel)."""
    from roi_align_pin import check_roi_align_against_golden
    g = gold("roi_align_fwd")
    feat, rois = syn.roi_align_golden_inputs(C, H, W, B)
    ft, rt = torch.from_numpy(feat).to(DEV), torch.from_numpy(rois).to(DEV)
    for nhwc_in in (True, False):
        f = ft.contiguous(memory_format=torch.channels_last) if nhwc_in else ft
        for out_nchw in (True, False):
            a8 = ops.roi_align(f, rt, 8, 8, 1.0 / 16.0, avg=False, out_nchw=out_nchw)
            a7 = ops.roi_align(f, rt, 7, 7, 1.0 / 16.0, avg=True, out_nchw=out_nchw)
            p7 = ops.roi_align(f, rt, 7, 7, 1.0 / 16.0, avg=False, out_nchw=out_nchw)
            check_roi_align_against_golden(g, C, *(t.contiguous().cpu().numpy() for t in (a8, a7, p7)))


@pytest.mark.parametrize("B,C,H,W,PH,PW", [(2, 128, 9, 11, 7, 7), (1, 256, 19, 32, 7, 7), (2, 8, 9, 11, 7, 7), (2, 128, 19, 32, 14, 7),
                                           (1, 128, 9, 11, 3, 5)])
@pytest.mark.parametrize("avg", [True, False])
def test_roi_align_bwd_is_the_transpose_of_the_pinned_forward(ops, oracle, monkeypatch, B, C, H, W, PH, PW, avg):
    """Both HIP backwards (the deterministic gather where it applies: NHWC, C % 128 == 0; and the atomic scatter) against
    F^T g computed in float64 from the coefficients of the PINNED forward (tests/roi_align_pin.py): the backward has no
    reference to run (roi_align.c:175), its definition as the forward's transpose is what pins it."""
    from i2vsgg_amd import ops as O
    from roi_align_pin import roi_align_matrix, roi_align_bwd_from_matrix
    cops, _ = oracle
    rng = np.random.default_rng(B * 100 + H + C)
    rois = syn.roi_cases(31 + H, B, H, W)
    mats = roi_align_matrix(rois, B, H, W, PH + int(avg), PW + int(avg), 1 / 16.0, cops.roi_align_fwd)
    gout = rng.standard_normal((rois.shape[0], C, PH, PW), dtype=np.float32)      # non-square grids too (14 x 7: 15 sample rows)
    want = roi_align_bwd_from_matrix(mats, cops.avgpool2x2_bwd(gout) if avg else gout, rois, B, C, H, W)
    scale = np.abs(want).max()
    rt = torch.from_numpy(rois).to(DEV)
    for gather in (True, False):
        monkeypatch.setattr(O, "ROIALIGN_BWD_GATHER", gather)
        for nhwc in (True, False):
            feat = torch.zeros((B, C, H, W), device=DEV)
            feat = (feat.contiguous(memory_format=torch.channels_last) if nhwc else feat).requires_grad_()
            g = torch.from_numpy(gout).to(DEV)
            O.roi_align(feat, rt, PH, PW, 1.0 / 16.0, avg=avg, out_nchw=not nhwc).backward(
                g if not nhwc else g.contiguous(memory_format=torch.channels_last))
            err = np.abs(feat.grad.cpu().numpy() - want).max()
            assert err <= 2e-6 * scale, (gather, nhwc, err / scale)


@pytest.mark.parametrize("C,H,W,B", [(4, 9, 11, 2), (64, 19, 32, 2), (1024, 38, 63, 1)])
@pytest.mark.parametrize("avg", [True, False])
def test_roi_align_fwd_bit_exact(ops, oracle, C, H, W, B, avg):
    cops, _ = oracle
    rng = np.random.default_rng(C + H)
    feat = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rois = _rois_cases(rng, B, H, W)
    ph = pw = 7
    if avg:
        ref = cops.roi_align_avg_fwd(feat, rois, ph, pw, 1.0 / 16.0)
    else:
        ref = cops.roi_align_fwd(feat, rois, ph, pw, 1.0 / 16.0)
    ft = torch.from_numpy(feat).to(DEV)
    rt = torch.from_numpy(rois).to(DEV)
    for nhwc_in in (True, False):
        for out_nchw in (True, False):
            f = ft.contiguous(memory_format=torch.channels_last) if nhwc_in else ft
            out = ops.roi_align(f, rt, ph, pw, 1.0 / 16.0, avg=avg, out_nchw=out_nchw)
            assert out.shape == ref.shape
            got = out.cpu().numpy()
            assert np.array_equal(got, ref), "nhwc_in=%s out_nchw=%s maxdiff=%g" % (
                nhwc_in, out_nchw, np.abs(got - ref).max())


def test_roi_align_full_image_last_row_is_zero(ops):
    """Property of the legacy grid: a full-image ROI samples row/col H, W -> 0."""
    feat = torch.ones((1, 8, 38, 63), device=DEV).contiguous(memory_format=torch.channels_last)
    rois = torch.tensor([[0, 0, 0, 63 * 16 - 1, 38 * 16 - 1]], device=DEV, dtype=torch.float32)
    out = ops.roi_align(feat, rois, 8, 8, 1.0 / 16.0, avg=False)
    assert torch.all(out[0, :, 7, :] == 0) and torch.all(out[0, :, :, 7] == 0)
    assert torch.all(out[0, :, :7, :7] == 1)


@pytest.mark.parametrize("C,H,W,B", [(8, 9, 11, 2), (256, 19, 32, 2)])
@pytest.mark.parametrize("avg", [True, False])
def test_roi_align_bwd(ops, oracle, C, H, W, B, avg):
    cops, _ = oracle
    rng = np.random.default_rng(7 + C)
    rois = _rois_cases(rng, B, H, W)
    gout = rng.standard_normal((rois.shape[0], C, 7, 7), dtype=np.float32)
    ref = (cops.roi_align_avg_bwd if avg else cops.roi_align_bwd)(gout, rois, (B, C, H, W), 1.0 / 16.0)
    for nhwc in (True, False):
        feat = torch.zeros((B, C, H, W), device=DEV, requires_grad=True)
        f = feat.contiguous(memory_format=torch.channels_last) if nhwc else feat
        out = ops.roi_align(f, torch.from_numpy(rois).to(DEV), 7, 7, 1.0 / 16.0, avg=avg, out_nchw=not nhwc)
        out.backward(torch.from_numpy(gout).to(DEV))
        # atomics: summation order is free -> tolerance, not bit-exactness
        np.testing.assert_allclose(feat.grad.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("C,H,W,B,R", [(1024, 38, 63, 4, 32), (128, 19, 32, 2, 9), (256, 50, 67, 1, 40), (128, 19, 32, 2, 300)])
@pytest.mark.parametrize("avg", [True, False])
def test_roi_align_bwd_gather_is_deterministic_and_equals_the_scatter(ops, oracle, monkeypatch, C, H, W, B, R, avg):
    """Round 4: the backward of RoIAlign(Avg) as a gather (i2v_roi_align_bwd_gather: NHWC in and out, C % 128 == 0) -- every
    element of the gradient map written once, contributions summed in the serial order of roi_align_kernel.cu:94-143 run as a
    loop (roi, sample row, sample column ascending), no atomics, no zero-fill.  Against the C restatement of that loop, against
    the atomic scatter it replaces, and twice for the same bits; 4 frames x 32 ROIs at full size (8192 samples: four list
    passes per workgroup), ROIs of every size incl. sub-cell ones (all 64 samples of a ROI on one map row) and ones outside;
    2 x 300 ROIs on a small map (five list passes, lists of more than a hundred pairs per row: several record chunks)."""
    from i2vsgg_amd import ops as O
    cops, _ = oracle
    rng = np.random.default_rng(C + H + R)
    rois = np.concatenate([_rois_cases(rng, B, H, W)] + [np.concatenate([np.full((R, 1), b, np.float32),
                          syn.boxes(R * 7 + b, R, H * 16, W * 16, 8, min(H, W) * 12)], 1) for b in range(B)]).astype(np.float32)
    gout = rng.standard_normal((rois.shape[0], C, 7, 7), dtype=np.float32)
    ref = (cops.roi_align_avg_bwd if avg else cops.roi_align_bwd)(gout, rois, (B, C, H, W), 1.0 / 16.0)
    rt, gt = torch.from_numpy(rois).to(DEV), torch.from_numpy(gout).to(DEV).contiguous(memory_format=torch.channels_last)

    def run():
        feat = torch.zeros((B, C, H, W), device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        O.roi_align(feat, rt, 7, 7, 1.0 / 16.0, avg=avg).backward(gt)
        return feat.grad

    assert O.ROIALIGN_BWD_GATHER
    g1, g2 = run(), run()
    assert torch.equal(g1, g2)                                           # no atomics: the same bits every time
    # round 5's form of the gather (I2V_TUNE_ROIALIGN_BWD = 0: lane = channel x cell parity, staged batches) against round 6's
    # (waves that own channel slices, 16-byte read-add-writes, pooled columns loaded once): the same sums in another order
    from i2vsgg_amd._lib import TUNE, lib
    assert lib.i2v_get_tuning(TUNE["I2V_ROIALIGN_BWD"]) == 1 and lib.i2v_set_tuning(TUNE["I2V_ROIALIGN_BWD"], 0) == 0
    try:
        g5, g5b = run(), run()
    finally:
        lib.i2v_set_tuning(TUNE["I2V_ROIALIGN_BWD"], 1)
    assert torch.equal(g5, g5b) and float((g1 - g5).abs().max()) <= 2e-6 * float(np.abs(ref).max())
    monkeypatch.setattr(O, "ROIALIGN_BWD_GATHER", False)
    sc = run()                                                           # the atomic scatter
    scale = float(np.abs(ref).max())
    assert float((g1 - sc).abs().max()) <= 2e-6 * scale
    np.testing.assert_allclose(g1.cpu().numpy(), ref, rtol=0, atol=2e-6 * scale)
    assert float(g1.abs().sum()) > 0


@pytest.mark.parametrize("C,H,W,B", [(4, 9, 11, 2), (6, 9, 11, 2), (64, 19, 32, 2), (1024, 38, 63, 1)])
@pytest.mark.parametrize("sampling", [0, 2])
def test_roi_align_sampled_fwd_bwd(ops, oracle, C, H, W, B, sampling):
    """roi_layers.ROIAlign (sampling grid): forward bit-exact against the C restatement for both map layouts and both
    output layouts; backward within the tolerance fp32 atomics allow; a roi of another batch gives zeros."""
    cops, _ = oracle
    rng = np.random.default_rng(3 * C + H + sampling)
    feat = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rois = _rois_cases(rng, B, H, W)
    ref = cops.roi_align_sampled_fwd(feat, rois, 7, 7, 1.0 / 16.0, sampling)
    gout = rng.standard_normal(ref.shape, dtype=np.float32)
    refg = cops.roi_align_sampled_bwd(gout, rois, feat.shape, 1.0 / 16.0, sampling)
    rt = torch.from_numpy(rois).to(DEV)
    for nhwc in (True, False):
        for out_nchw in (True, False):
            ft = torch.from_numpy(feat).to(DEV).requires_grad_()
            f = ft.contiguous(memory_format=torch.channels_last) if nhwc else ft
            out = ops.roi_align_sampled(f, rt, 7, 7, 1.0 / 16.0, sampling, out_nchw=out_nchw)
            got = out.detach().cpu().numpy()
            assert np.array_equal(got, ref), "nhwc=%s out_nchw=%s maxdiff=%g" % (nhwc, out_nchw, np.abs(got - ref).max())
            out.backward(torch.from_numpy(gout).to(DEV))
            np.testing.assert_allclose(ft.grad.cpu().numpy(), refg, rtol=2e-5, atol=2e-5)
    bad = rt.clone()
    bad[:, 0] = B + 3
    assert torch.all(ops.roi_align_sampled(torch.from_numpy(feat).to(DEV), bad, 7, 7, 1.0 / 16.0, sampling) == 0)


def test_roi_layers_roialign_module(ops, oracle):
    """The module with the reference's constructor (faster_rcnn_SGG_emb.py:47) and an affine map: every bin returns
    the map at its centre."""
    from i2vsgg_amd.model.roi_layers import ROIAlign
    cops, _ = oracle
    H, W = 38, 63
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    feat = torch.stack([0.5 * yy + 0.25 * xx, 1.0 - 0.125 * yy + 0.5 * xx])[None].to(DEV)
    rois = torch.tensor([[0, 32.0, 48.0, 480.0, 400.0]], device=DEV)
    m = ROIAlign((7, 7), 1.0 / 16.0, 0, out_nchw=True)
    out = m(feat, rois)
    cy = 3.0 + (torch.arange(7) + 0.5) * (22.0 / 7)
    cx = 2.0 + (torch.arange(7) + 0.5) * (28.0 / 7)
    want = 0.5 * cy[:, None] + 0.25 * cx[None, :]
    assert torch.allclose(out[0, 0].cpu(), want, atol=2e-5)
    assert np.array_equal(out.cpu().numpy(), cops.roi_align_sampled_fwd(feat.cpu().numpy(), rois.cpu().numpy(), 7, 7, 1 / 16.0, 0))
    assert "sampling_ratio=0" in repr(m)


@pytest.mark.parametrize("C,H,W,B", [(4, 9, 11, 2), (100, 19, 32, 2), (1024, 38, 63, 1)])
def test_roi_pool_fwd_bwd(ops, oracle, C, H, W, B):
    cops, _ = oracle
    rng = np.random.default_rng(11 + C)
    feat = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rois = _rois_cases(rng, B, H, W)
    ref, refarg = cops.roi_pool_fwd(feat, rois, 7, 7, 1.0 / 16.0)
    gout = rng.standard_normal(ref.shape, dtype=np.float32)
    refg = cops.roi_pool_bwd(gout, rois, refarg, feat.shape)
    for nhwc in (True, False):
        for out_nchw in (True, False):
            ft = torch.from_numpy(feat).to(DEV).requires_grad_()
            f = ft.contiguous(memory_format=torch.channels_last) if nhwc else ft
            out, arg = ops.roi_pool(f, torch.from_numpy(rois).to(DEV), 7, 7, 1.0 / 16.0, out_nchw=out_nchw,
                                    return_argmax=True)
            assert np.array_equal(out.detach().cpu().numpy(), ref)
            assert np.array_equal(arg.cpu().numpy(), refarg)
            out.backward(torch.from_numpy(gout).to(DEV))
            np.testing.assert_allclose(ft.grad.cpu().numpy(), refg, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 1000, 6000, 12000])
def test_nms_keep_matches_reference_golden(ops, gold, n):
    """The 36 keep lists of the reference's nms_cpu (uniform and clustered boxes, thresholds 0.7 / 0.3), bit for bit, in every
    form of the scan (I2V_TUNE_NMS_SCAN): 2 = round 6's default (1024-row super-blocks resolved by fixed-point sweeps), 3 = the
    same with three sweeps at most (clustered super-blocks fall back to the serial resolve: both paths in one launch),
    1 = super-blocks with the serial resolve, 0 = round 5's 64-row scan."""
    from i2vsgg_amd._lib import TUNE, lib
    g = gold("nms_keep")
    assert lib.i2v_get_tuning(TUNE["I2V_NMS_SCAN"]) == 2
    try:
        for mode in (2, 3, 1, 0):
            assert lib.i2v_set_tuning(TUNE["I2V_NMS_SCAN"], mode) == 0
            for clustered in (False, True):
                dets = syn.tie_free_dets(1000 + n, n, clustered=clustered)
                dt = torch.from_numpy(dets).to(DEV)
                for th in (0.7, 0.3):
                    ref = g["n%d_%s_t%02d" % (n, "c" if clustered else "u", int(th * 10))]
                    keep, num = ops.nms_sorted(dt, th)
                    k = int(num.item())
                    assert k == ref.size, (mode, clustered, th)
                    assert np.array_equal(keep[0, :k].cpu().numpy(), ref), (mode, clustered, th)
    finally:
        lib.i2v_set_tuning(TUNE["I2V_NMS_SCAN"], 2)


def test_nms_batched_early_exit_and_empty(ops, oracle):
    cops, _ = oracle
    dets = np.stack([syn.tie_free_dets(77 + i, 3000, clustered=bool(i % 2)) for i in range(3)])
    from i2vsgg_amd._lib import TUNE, lib
    try:
        for mode in (2, 3, 1, 0):                         # (every form of the scan: test_nms_keep_matches_reference_golden)
            lib.i2v_set_tuning(TUNE["I2V_NMS_SCAN"], mode)
            for cap in (300, 1100, 17):                   # the cut inside the first super-block, inside the second, inside the first rows
                keep, num = ops.nms_sorted(torch.from_numpy(dets).to(DEV), 0.7, max_keep=cap)
                for i in range(3):
                    ref = cops.nms_sorted(dets[i], 0.7)[:cap]
                    assert int(num[i]) == ref.size, (mode, cap, i)
                    assert np.array_equal(keep[i, :ref.size].cpu().numpy(), ref), (mode, cap, i)
    finally:
        lib.i2v_set_tuning(TUNE["I2V_NMS_SCAN"], 2)
    keep, num = ops.nms_sorted(torch.zeros((0, 5), device=DEV), 0.7)
    assert int(num.item()) == 0


@pytest.mark.parametrize("n", [5, 4096, 4097, 21546, 40000])
def test_sort_desc(ops, n):
    rng = np.random.default_rng(n)
    keys = rng.standard_normal((2, n)).astype(np.float32)
    keys[0, : n // 3] = keys[0, n // 3: 2 * (n // 3)][: n // 3]      # force ties
    order = ops.sort_desc(torch.from_numpy(keys).to(DEV)).cpu().numpy()
    for s in range(2):
        ref = np.argsort(-keys[s], kind="stable")
        assert np.array_equal(order[s], ref)


def _rpn_inputs(seed, B, H=38, W=63):
    rng = np.random.default_rng(seed)
    n = B * 9 * H * W
    fg = ((rng.permutation(n).astype(np.float32) + 1.0) / np.float32(n + 2)).reshape(B, 9, H, W)
    deltas = (rng.standard_normal((B, 36, H, W)) * 0.25).astype(np.float32)
    return fg, deltas


def test_rpn_decode_bit_exact_vs_oracle_and_reference(ops, oracle, gold):
    _, rpn = oracle
    g = gold("decode_clip")
    rng = np.random.default_rng(101)
    anc = rpn.anchor_grid(38, 63)
    deltas = (rng.standard_normal((2, anc.shape[0], 4)) * np.array([0.3, 0.3, 0.5, 0.5])).astype(np.float32)
    # (B,N,4) in (y,x,a) order -> NCHW bbox map (B,36,H,W)
    bbox = deltas.reshape(2, 38, 63, 36).transpose(0, 3, 1, 2).copy()
    cls = np.zeros((2, 18, 38, 63), np.float32)
    base = torch.from_numpy(rpn.base_anchors().astype(np.float32)).to(DEV)
    prop, _ = ops.rpn_decode(torch.from_numpy(cls).to(DEV), torch.from_numpy(bbox).to(DEV),
                             torch.from_numpy(g["im_info"]).to(DEV), base, 16, is_prob=True)
    got = prop.cpu().numpy()
    for b in range(2):
        mine = rpn.decode_clip(anc, deltas[b], g["im_info"][b, 0], g["im_info"][b, 1])
        assert np.array_equal(got[b], mine)                     # bit-exact vs the oracle
        np.testing.assert_allclose(got[b], g["proposals"][b], rtol=3e-7, atol=2e-4)   # exp ulp vs torch


@pytest.mark.parametrize("B", [1, 2])
def test_rpn_proposal_layer_vs_reference_golden(ops, oracle, gold, B):
    _, rpn = oracle
    g = gold("proposal_layer")
    fg, deltas = _rpn_inputs(200 + B, B)
    prob = np.concatenate([1.0 - fg, fg], 1).astype(np.float32)
    info = np.array([[600, 1000, 1.0]] * B, np.float32)
    base = torch.from_numpy(rpn.base_anchors().astype(np.float32)).to(DEV)
    for mode, pre, post in (("train", 12000, 2000), ("test", 6000, 300), ("target", 12000, 32)):
        rois, kept, num = ops.rpn_proposal(torch.from_numpy(prob).to(DEV), torch.from_numpy(deltas).to(DEV),
                                           torch.from_numpy(info).to(DEV), base, 16, pre, post, 0.7, is_prob=True,
                                           want_index=True)
        ref_rois, ref_kept = rpn.proposal_layer(fg, deltas, info, pre, post, 0.7)
        got = rois.cpu().numpy()
        assert np.array_equal(got, ref_rois)                    # bit-exact vs the oracle
        for b in range(B):                                      # identical kept anchor indices
            k = int(num[b])
            assert k == ref_kept[b].size
            assert np.array_equal(kept[b, :k].cpu().numpy(), ref_kept[b])
        np.testing.assert_allclose(got, g["rois_B%d_%s" % (B, mode)], rtol=3e-7, atol=2e-4)


def test_rpn_proposal_from_logits(ops, oracle):
    """is_prob=0: the pairwise softmax of rpn.py:69-71 is fused into the decode kernel."""
    _, rpn = oracle
    rng = np.random.default_rng(5)
    cls = rng.standard_normal((1, 18, 38, 63)).astype(np.float32)
    _, deltas = _rpn_inputs(6, 1)
    prob = F.softmax(torch.from_numpy(cls).view(1, 2, 9 * 38, 63), 1).view(1, 18, 38, 63).numpy()
    info = np.array([[600, 1000, 1.0]], np.float32)
    base = torch.from_numpy(rpn.base_anchors().astype(np.float32)).to(DEV)
    _, score = ops.rpn_decode(torch.from_numpy(cls).to(DEV), torch.from_numpy(deltas).to(DEV),
                              torch.from_numpy(info).to(DEV), base, 16, is_prob=False)
    ref = prob[0, 9:].transpose(1, 2, 0).reshape(-1)
    np.testing.assert_allclose(score[0].cpu().numpy(), ref, rtol=2e-6, atol=1e-7)


def test_bbox_overlaps_vs_reference_golden(ops, oracle, gold):
    _, rpn = oracle
    g = gold("box_math")
    ov, mx, am = ops.bbox_overlaps(torch.from_numpy(g["rois"]).to(DEV), torch.from_numpy(g["gt"]).to(DEV),
                                   want_matrix=True)
    assert np.array_equal(ov.cpu().numpy(), g["overlaps_rois"])
    assert np.array_equal(mx.cpu().numpy(), g["overlaps_rois"].max(2))
    assert np.array_equal(am.cpu().numpy(), g["overlaps_rois"].argmax(2))
    anc = torch.from_numpy(rpn.anchor_grid(38, 63)[::7].copy()).to(DEV)
    ov, _, _ = ops.bbox_overlaps(anc, torch.from_numpy(g["gt"]).to(DEV), want_matrix=True)
    assert np.array_equal(ov.cpu().numpy(), g["overlaps_anchors"])


# --------------------------------------------------------------------------- conv / linear
CONV_CASES = [
    # B, Cin, H, W, Cout, K, stride, pad
    (2, 64, 20, 27, 64, 1, 1, 0),
    (2, 64, 20, 27, 256, 1, 1, 0),
    (1, 128, 19, 33, 128, 3, 1, 1),
    (2, 256, 21, 30, 512, 1, 2, 0),
    (1, 4, 67, 91, 64, 7, 2, 3),
    (1, 1024, 38, 63, 512, 3, 1, 1),     # RPN_Conv shape (split-free, 64x64 tiles)
    (3, 512, 7, 7, 18, 1, 1, 0),         # Cout not a multiple of 4
    (1, 96, 16, 16, 128, 5, 2, 2),       # vrd.conv_lo.1 (Cin not a power of two)
    (4, 128, 8, 8, 64, 8, 1, 0),         # vrd.conv_lo.2 -> 1x1 output, deep K (split-K)
    # strided KxK: the dgrad runs one sub-filter correlation per input-pixel parity class
    (2, 32, 17, 23, 64, 3, 2, 1),        # odd sizes: the last row/column has no window of its own
    (1, 32, 16, 21, 32, 5, 2, 2),
    (2, 16, 12, 10, 32, 2, 2, 0),        # sub-filters are single taps
    (1, 16, 13, 14, 16, 3, 3, 0),
    (1, 16, 14, 15, 32, 4, 2, 1),
    (1, 16, 11, 11, 16, 2, 3, 0),        # filter smaller than the stride: one parity class receives nothing
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_vs_torch_fp32(ops, case):
    B, Cin, H, W, Cout, K, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, K, K), dtype=np.float32) / np.sqrt(Cin * K * K)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    sh = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    xt, wt = torch.from_numpy(x), torch.from_numpy(w)
    ref = F.conv2d(xt, wt, stride=s, padding=p)
    res = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    xd, wd = xt.to(DEV), wt.to(DEV)
    scd, shd, resd = torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV), res.to(DEV)
    tol = dict(rtol=2e-5, atol=2e-5)
    y = ops.conv2d(xd, wd, stride=s, pad=p)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), **tol)
    y = ops.conv2d(xd, wd, scd, shd, resd, s, p, relu=True)
    r2 = F.relu(ref * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1) + res)
    np.testing.assert_allclose(y.cpu().numpy(), r2.numpy(), **tol)
    y = ops.conv2d(xd, wd, None, shd, None, s, p, relu=False)
    np.testing.assert_allclose(y.cpu().numpy(), (ref + torch.from_numpy(sh).view(1, -1, 1, 1)).numpy(), **tol)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] != 7])
def test_conv_bwd_vs_torch_autograd(ops, case):
    B, Cin, H, W, Cout, K, s, p = case
    rng = np.random.default_rng(sum(case) + 1)
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, K, K), dtype=np.float32) / np.sqrt(Cin * K * K)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_() for a in (x, w, b))
    ref = F.relu(F.conv2d(xt, wt, bt, stride=s, padding=p))
    gy = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    ref.backward(gy)
    xd, wd, bd = (torch.from_numpy(a).to(DEV).requires_grad_() for a in (x, w, b))
    y = ops.conv2d(xd, wd, None, bd, None, s, p, relu=True)
    y.backward(gy.to(DEV))
    tol = dict(rtol=3e-4, atol=3e-4)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), **tol)
    # eligible 3x3 layers take the Winograd F(4x4,3x3) filter gradient: fp32 error ~2e-5 of the tensor's range (the
    # transform constants reach 8 and 1/24), where the direct kernel has ~1e-6
    wtol = dict(rtol=3e-4, atol=max(3e-4, 5e-5 * float(wt.grad.abs().max())))
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wt.grad.numpy(), **wtol)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), bt.grad.numpy(), **tol)


def test_conv_bn_residual_backward(ops):
    """Bottleneck-style epilogue: frozen-BN scale/shift + residual + ReLU; grads to x, w and res."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 64, 12, 15), dtype=np.float32)
    w = (rng.standard_normal((128, 64, 1, 1), dtype=np.float32) / 8).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    sh = rng.uniform(-0.5, 0.5, 128).astype(np.float32)
    r = rng.standard_normal((2, 128, 12, 15), dtype=np.float32)
    xt, wt, rt = (torch.from_numpy(a).requires_grad_() for a in (x, w, r))
    ref = F.relu(F.conv2d(xt, wt) * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1) + rt)
    gy = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    ref.backward(gy)
    xd, wd, rd = (torch.from_numpy(a).to(DEV).requires_grad_() for a in (x, w, r))
    y = ops.conv2d(xd, wd, torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV), rd, 1, 0, relu=True)
    y.backward(gy.to(DEV))
    tol = dict(rtol=3e-4, atol=3e-4)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), **tol)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wt.grad.numpy(), **tol)
    np.testing.assert_allclose(rd.grad.cpu().numpy(), rt.grad.numpy(), **tol)


# the last three are tall-and-narrow: the epilogue backward's row-lane reduction (N/4 column groups x row lanes)
@pytest.mark.parametrize("B,C,N,H,W", [(2, 256, 256, 38, 63), (3, 64, 128, 9, 7), (5, 512, 512, 4, 4)])
def test_trained_3x3_winograd_fwd_and_dgrad_vs_torch_autograd(ops, B, C, N, H, W):
    """A stride-1 / pad-1 3x3 layer under autograd runs forward and data gradient as Winograd F(4x4,3x3) (wgrad stays
    direct): output, input gradient and filter gradient against torch's fp32 conv in float64."""
    assert ops.WINOGRAD_TRAIN
    g = torch.Generator(device="cpu").manual_seed(B * C + H)
    x = torch.randn(B, C, H, W, generator=g).to(DEV).requires_grad_()
    w = (torch.randn(N, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(DEV).requires_grad_()
    sc, sh = (torch.rand(N, generator=g) + 0.5).to(DEV), (torch.rand(N, generator=g) - 0.5).to(DEV)
    gy = torch.randn(B, N, H, W, generator=g).to(DEV)
    y = ops.conv2d(x, w, sc, sh, None, 1, 1, relu=True, winograd=True)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    pre = F.conv2d(xd, wd, padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    assert rel(y, torch.relu(pre.detach())) < 1e-4
    # the ReLU mask is the kernel's own: an output within ~1e-5 of zero may sit on the other side of the ReLU in fp32,
    # and one flipped element moves the input gradient by a whole tap -- masks must agree everywhere else
    mask = y.detach() > 0
    flips = (mask != (pre.detach() > 0))
    assert not bool((flips & (pre.detach().abs() > 1e-4 * pre.detach().abs().max())).any())
    (pre * mask * gy.double()).sum().backward()
    assert rel(x.grad, xd.grad) < 1e-4 and rel(w.grad, wd.grad) < 1e-4
    # the same layer with the direct kernels (the default of ops.conv2d)
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    y2 = ops.conv2d(x2, w2, sc, sh, None, 1, 1, relu=True)
    assert rel(y, y2.double()) < 1e-4 and not torch.equal(y, y2)


@pytest.mark.parametrize("M,K,N", [(8, 50176, 64), (128, 4096, 300), (32, 600, 256), (62, 300, 1024), (5, 64, 1),
                                   (16384, 100, 96), (4099, 64, 128), (1024, 32, 512)])
def test_linear_fwd_bwd(ops, M, K, N):
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal((N,), dtype=np.float32)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_() for a in (x, w, b))
    ref = F.relu(F.linear(xt, wt, bt))
    gy = torch.from_numpy(rng.standard_normal((M, N), dtype=np.float32))
    ref.backward(gy)
    xd, wd, bd = (torch.from_numpy(a).to(DEV).requires_grad_() for a in (x, w, b))
    y = ops.linear(xd, wd, bd, relu=True)
    y.backward(gy.to(DEV))
    tol = dict(rtol=3e-4, atol=3e-4)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), **tol)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), **tol)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wt.grad.numpy(), **tol)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), bt.grad.numpy(), **tol)


@pytest.mark.parametrize("H,W", [(300, 500), (49, 66), (50, 65)])
def test_maxpool_ceil_mode(ops, H, W):
    x = torch.from_numpy(np.random.default_rng(H).standard_normal((2, 64, H, W), dtype=np.float32))
    ref = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
    y = ops.maxpool3x3s2(x.to(DEV))
    assert y.shape == ref.shape
    assert torch.equal(y.cpu(), ref)


def test_sgd_momentum_matches_torch(ops):
    rng = np.random.default_rng(9)
    p0 = rng.standard_normal(100003, dtype=np.float32)
    pt = torch.from_numpy(p0.copy()).requires_grad_()
    opt = torch.optim.SGD([pt], lr=0.01, momentum=0.9, weight_decay=5e-4)
    pd = torch.from_numpy(p0.copy()).to(DEV)
    md = torch.zeros_like(pd)
    for it in range(3):
        g = rng.standard_normal(100003, dtype=np.float32)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.sgd_momentum_(pd, torch.from_numpy(g).to(DEV), md, 0.01, 0.9, 5e-4)
    np.testing.assert_allclose(pd.cpu().numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_sgd_momentum_multi_equals_per_tensor(ops):
    """One launch for many small tensors == one launch per tensor, bitwise (60 tensors: two table chunks)."""
    rng = np.random.default_rng(10)
    sizes = [1, 3, 300, 4096, 4097, 65536 + 5, 256 * 16, 256 * 16 + 1] + [int(v) for v in rng.integers(1, 20000, 52)]
    mk = lambda n: torch.from_numpy(rng.standard_normal(n, dtype=np.float32)).to(DEV)
    ps, gs, ms = [mk(n) for n in sizes], [mk(n) for n in sizes], [mk(n) for n in sizes]
    lrs = [0.01 * (1 + (i % 2)) for i in range(len(sizes))]
    wds = [5e-4 * (i % 3 != 0) for i in range(len(sizes))]
    p2, m2 = [t.clone() for t in ps], [t.clone() for t in ms]
    for p, g, m, lr, wd in zip(ps, gs, ms, lrs, wds):
        ops.sgd_momentum_(p, g, m, lr, 0.9, wd)
    ops.sgd_momentum_multi_(p2, gs, m2, lrs, wds, 0.9)
    for a, b, c, d in zip(ps, p2, ms, m2):
        assert torch.equal(a, b) and torch.equal(c, d)


def test_dstyle_pool(ops):
    rng = np.random.default_rng(4)
    x1 = torch.from_numpy(rng.standard_normal((2, 300, 2560), dtype=np.float32)).requires_grad_()
    x2 = torch.from_numpy(rng.standard_normal((2, 300, 2560), dtype=np.float32)).requires_grad_()
    ref = (x1 * x2).reshape(2, 300, 512, 5).sum(-1).sum(1)
    gz = torch.from_numpy(rng.standard_normal((2, 512), dtype=np.float32))
    ref.backward(gz)
    d1, d2 = (t.detach().to(DEV).requires_grad_() for t in (x1, x2))
    z = ops.dstyle_pool(d1, d2, 512, 5)
    z.backward(gz.to(DEV))
    np.testing.assert_allclose(z.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(d1.grad.cpu().numpy(), x1.grad.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d2.grad.cpu().numpy(), x2.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_cpu_tensor_is_rejected_loudly(ops):
    with pytest.raises(Exception):
        ops.roi_align(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 7, 7, 1 / 16.0)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_conv_every_tile_shape(ops, tile):
    """Each (BM x BN) instantiation of conv_igemm_f32 forced through the tuning hook, incl. ragged M/N/K edges, residual and
    split-K.  (Rounds 1-5 had a second, 8-wave loader / MFMA form per tile behind an experiments build: gone in round 6 -- the
    hook refuses its bits.)"""
    from i2vsgg_amd import _lib
    assert _lib.lib.i2v_conv_set_tile(tile | (2 << 8)) != 0 and b"experiment" in _lib.lib.i2v_last_error()
    rng = np.random.default_rng(tile)
    x = rng.standard_normal((2, 72, 13, 17), dtype=np.float32)          # M = 442 (ragged), K = 648 (not /32)
    w = (rng.standard_normal((100, 72, 3, 3), dtype=np.float32) / 25).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1)
    res = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    x8 = rng.standard_normal((1, 512, 5, 6), dtype=np.float32)                       # M = 30, deep K -> split-K
    w8 = (rng.standard_normal((64, 512, 3, 3), dtype=np.float32) / 60).astype(np.float32)
    ref8 = F.conv2d(torch.from_numpy(x8), torch.from_numpy(w8), padding=1)
    assert _lib.lib.i2v_conv_set_tile(tile) == 0
    try:
        y = ops.conv2d(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), pad=1)
        y2 = ops.conv2d(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), None, None, res.to(DEV), 1, 1, relu=True)
        y8 = ops.conv2d(torch.from_numpy(x8).to(DEV), torch.from_numpy(w8).to(DEV), pad=1)
    finally:
        _lib.lib.i2v_conv_set_tile(-1)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y2.cpu().numpy(), F.relu(ref + res).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y8.cpu().numpy(), ref8.numpy(), rtol=2e-5, atol=2e-5)


# --------------------------------------------------------------------------- edge cases
def test_empty_and_degenerate_inputs(ops):
    feat = torch.randn(1, 64, 10, 12, device=DEV).contiguous(memory_format=torch.channels_last)
    empty = torch.zeros((0, 5), device=DEV)
    assert ops.roi_align(feat, empty, 7, 7, 1 / 16.0).shape == (0, 64, 7, 7)          # R = 0
    assert ops.roi_pool(feat, empty, 7, 7, 1 / 16.0).shape == (0, 64, 7, 7)
    with pytest.raises(ValueError):
        ops.roi_align(feat, torch.zeros((3, 4), device=DEV), 7, 7, 1 / 16.0)            # roi_align.c:28-31: (K,5) only
    # a single box: NMS keeps it; two identical boxes: the second is suppressed
    one = torch.tensor([[10., 10., 50., 50., 0.9]], device=DEV)
    keep, num = ops.nms_sorted(one, 0.7)
    assert int(num) == 1 and int(keep[0, 0]) == 0
    two = torch.tensor([[10., 10., 50., 50., 0.9], [10., 10., 50., 50., 0.8]], device=DEV)
    keep, num = ops.nms_sorted(two, 0.7)
    assert int(num) == 1
    # IoU exactly at the threshold is KEPT (nms_cpu.py:31 `ovr <= thresh`): boxes 0..9 and 0..19 wide -> IoU 0.5
    th = torch.tensor([[0., 0., 9., 9., 0.9], [0., 0., 19., 9., 0.8]], device=DEV)
    keep, num = ops.nms_sorted(th, 0.5)
    assert int(num) == 2


def test_nms_large_n_uses_generic_scan(ops, oracle):
    """n = 20000 (> 12288): the non-pipelined scan path; still bit-exact against the oracle."""
    cops, _ = oracle
    dets = syn.tie_free_dets(99, 20000, clustered=True)
    keep, num = ops.nms_sorted(torch.from_numpy(dets).to(DEV), 0.7)
    ref = cops.nms_sorted(dets, 0.7)
    assert int(num) == ref.size and np.array_equal(keep[0, :ref.size].cpu().numpy(), ref)


def test_proposal_layer_fewer_survivors_than_post_nms_is_zero_padded(ops, oracle):
    """All anchors decode to nearly the same box -> a handful survive NMS; the rest of the (B,post,5) block is
    zero with the image index in column 0 (proposal_layer.py:158-161)."""
    _, rpn = oracle
    cls = torch.randn(2, 18, 8, 9, device=DEV)
    box = torch.zeros(2, 36, 8, 9, device=DEV)
    box[:, 2::4] = -20.0          # exp(-20) ~ 0: every proposal collapses to its anchor centre (tiny boxes)
    box[:, 3::4] = -20.0
    info = torch.tensor([[128., 144., 1.0]] * 2, device=DEV)
    base = torch.from_numpy(rpn.base_anchors().astype(np.float32)).to(DEV)
    rois, kept, num = ops.rpn_proposal(cls, box, info, base, 16, 6000, 300, 0.7, is_prob=False, want_index=True)
    assert rois.shape == (2, 300, 5)
    for b in range(2):
        k = int(num[b])
        assert 0 < k <= 300
        assert torch.all(rois[b, :, 0] == b)
        assert torch.all(rois[b, k:, 1:] == 0) and torch.all(kept[b, k:] == -1)


@pytest.mark.parametrize("R,ctx", [(5, False), (5, True), (32, True), (1, False)])
def test_dpixel_fused_vs_torch(ops, R, ctx):
    """netD_pixel in one kernel per direction (GRL, 3 pointwise convs, ReLUs, sigmoid, context mean) vs plain torch
    fp32 autograd of resnet_instance_styleD_bilinear.py:38-83 + net_utils.py:52-61.  R*49 is not a multiple of the
    32-row tile for R = 5 and R = 1."""
    rng = np.random.default_rng(100 + R)
    lamb = 0.3
    x = rng.standard_normal((R, 1024, 7, 7), dtype=np.float32)
    w1 = (rng.standard_normal((512, 1024), dtype=np.float32) * 0.03).astype(np.float32)
    w2 = (rng.standard_normal((128, 512), dtype=np.float32) * 0.05).astype(np.float32)
    w3 = (rng.standard_normal((128,), dtype=np.float32) * 0.2).astype(np.float32)
    gd = rng.standard_normal((R, 1, 7, 7), dtype=np.float32)
    gf = rng.standard_normal((R, 128, 1, 1), dtype=np.float32)

    class GRL(torch.autograd.Function):
        @staticmethod
        def forward(c, t):
            return t.view_as(t)

        @staticmethod
        def backward(c, g):
            return g * -lamb

    xt, w1t, w2t, w3t = (torch.from_numpy(a).requires_grad_() for a in (x, w1, w2, w3))
    h = F.relu(F.conv2d(GRL.apply(xt), w1t.view(512, 1024, 1, 1)))
    h = F.relu(F.conv2d(h, w2t.view(128, 512, 1, 1)))
    d_ref = torch.sigmoid(F.conv2d(h, w3t.view(1, 128, 1, 1)))
    f_ref = h.mean((2, 3), keepdim=True)
    loss = (d_ref * torch.from_numpy(gd)).sum() + ((f_ref * torch.from_numpy(gf)).sum() if ctx else 0.0)
    loss.backward()

    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    w1d, w2d, w3d = (torch.from_numpy(a).to(DEV).requires_grad_() for a in (w1, w2, w3))
    rows = xd.permute(0, 2, 3, 1).reshape(R * 49, 1024)
    d, feat = ops.dpixel(rows, w1d, w2d, w3d, lamb, 49, ctx)
    d4 = d.view(R, 7, 7, 1).permute(0, 3, 1, 2)
    out = (d4 * torch.from_numpy(gd).to(DEV)).sum()
    if ctx:
        out = out + (feat.view(R, 128, 1, 1) * torch.from_numpy(gf).to(DEV)).sum()
    out.backward()
    tol = dict(rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(d4.detach().cpu().numpy(), d_ref.detach().numpy(), **tol)
    if ctx:
        np.testing.assert_allclose(feat.detach().cpu().numpy().reshape(R, 128, 1, 1), f_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), **tol)
    # filter gradients are sums over R*49 rows of O(1) products: compare against the tensor's scale (fp32 summation
    # order differs).  A pre-activation within rounding of zero can take the other side of the ReLU in one of the two
    # implementations, which moves one filter row by a visible amount: allow a handful of such elements.
    for got, ref in ((w1d.grad, w1t.grad), (w2d.grad, w2t.grad), (w3d.grad, w3t.grad)):
        g, r = got.cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
        err, scale = np.abs(g - r), np.abs(r).max()
        assert (err > 1e-4 * scale).mean() < 1e-3 and err.max() < 2e-3 * scale, (err.max(), scale)


@pytest.mark.parametrize("R,C,agnostic,thresh,maxdet", [(300, 16, False, 0.0, 100), (300, 16, True, 0.05, 100),
                                                        (37, 5, False, 0.3, 0), (300, 36, False, 0.0, 100),
                                                        (64, 3, False, 0.99, 100)])
def test_detection_postprocess_vs_oracle(ops, oracle, R, C, agnostic, thresh, maxdet):
    """the device pass == the numpy restatement of test_net_instance_styleD_bilinear.py:151-221
    (whose pieces bbox_transform_inv / clip_boxes / nms_cpu are pinned by reference goldens): same detections, same
    order, bit-equal boxes and scores."""
    from oracle import rpn as orpn
    rng = np.random.default_rng(R * 100 + C)
    im_h, im_w, scale = 600.0, 1000.0, 1.6
    xy = rng.uniform(0, 1, (R, 2)) * [im_w - 120, im_h - 120]
    wh = rng.uniform(16, 300, (R, 2))
    rois = np.concatenate([np.zeros((R, 1)), xy, np.minimum(xy + wh, [im_w - 1, im_h - 1])], 1).astype(np.float32)
    # clustered rois so that the NMS has work to do
    rois[R // 2:, 1:] = rois[:R - R // 2, 1:] + rng.uniform(-6, 6, (R - R // 2, 4)).astype(np.float32)
    logits = rng.standard_normal((R, C)).astype(np.float32) * 2
    prob = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    pred = (rng.standard_normal((R, 4 if agnostic else 4 * C)) * 0.5).astype(np.float32)
    stds, means = (0.1, 0.1, 0.2, 0.2), (0.0, 0.0, 0.0, 0.0)
    want = orpn.detection_postprocess(rois, prob, pred, im_h, im_w, scale, agnostic, stds, means, thresh, 0.3, maxdet)
    dets, counts = ops.detection_postprocess(*(torch.from_numpy(a).to(DEV) for a in (rois, prob, pred)), im_h, im_w, scale,
                                             agnostic, stds, means, thresh, 0.3, maxdet)
    counts = counts.cpu().numpy()
    dets = dets.cpu().numpy()
    assert counts[0] == 0
    assert sum(len(w) for w in want) > 0 or thresh > 0.9
    for j in range(1, C):
        assert counts[j] == len(want[j]), (j, counts[j], len(want[j]))
        assert np.array_equal(dets[j, :counts[j]], want[j]), j


@pytest.mark.parametrize("tag", ["c16", "c8_t05", "c16_cag"])
def test_detection_postprocess_vs_reference_golden(ops, gold, tag):
    """Row f1 against the reference itself: tests/golden/det_postprocess.npz is the loop of
    test_net_instance_styleD_bilinear.py:151-221 run with the reference's own bbox_transform_inv / clip_boxes / nms_cpu
    (tools/gen_golden.py, tier "direct").  The device pass gives the same detections per class in the same order, bit-equal scores,
    boxes within one ulp of torch's vectorised fp32 exp (the kernel rounds exp once from fp64, as test_rpn_decode documents)."""
    g = gold("det_postprocess")
    im_h, im_w, scale, agnostic, thresh, nms_t, maxdet = (float(v) for v in g[tag + "_args"])
    dets, counts = ops.detection_postprocess(*(torch.from_numpy(g[tag + k]).to(DEV) for k in ("_rois", "_prob", "_pred")), im_h, im_w,
                                             scale, bool(agnostic), (0.1, 0.1, 0.2, 0.2), (0.0, 0.0, 0.0, 0.0), thresh, nms_t, int(maxdet))
    counts, dets = counts.cpu().numpy(), dets.cpu().numpy()
    want_counts = g[tag + "_count"]
    assert np.array_equal(counts[:len(want_counts)], want_counts) and int(want_counts.sum()) == 100
    ends = np.cumsum(want_counts)
    for j in range(1, len(want_counts)):
        want = g[tag + "_dets"][ends[j] - want_counts[j]:ends[j]]
        got = dets[j, :counts[j]]
        assert np.array_equal(got[:, 4], want[:, 4]), j
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=3e-7, atol=1e-4)


@pytest.mark.parametrize("tag", ["b9", "b5", "b16", "b1"])
def test_detection_output_vs_reference_golden(gold, tag):
    """Row f3 against the reference itself: tests/golden/detection_output.npz is lib/utils.py:584-628 compiled from its own lines
    and run (tier "extracted").  eval.detection_output (i2v_relation_topk on the device) returns the same top-100 triplets in the
    same order with bit-equal confidences and boxes; a one-box frame gives five Nones."""
    from i2vsgg_amd import eval as ev
    g = gold("detection_output")
    vrd = {"bboxes": g[tag + "_bboxes"], "classes": g[tag + "_classes"], "scores": g[tag + "_scores"], "ixs": g[tag + "_ixs"],
           "ixo": g[tag + "_ixo"], "rel_score": torch.from_numpy(g[tag + "_rel_score"]).to(DEV)}
    got = ev.detection_output(vrd, 100)
    if tag + "_none" in g.files:
        assert got == (None,) * 5
        return
    rlp, conf, sub, obj, idx = got
    assert np.array_equal(idx, g[tag + "_idx"]) and np.array_equal(rlp, g[tag + "_rlp"])
    assert np.array_equal(conf, g[tag + "_conf"])
    assert np.array_equal(sub, g[tag + "_sub"]) and np.array_equal(obj, g[tag + "_obj"])


@pytest.mark.parametrize("H,W,target,flip,rgb", [(375, 500, 600, False, True), (480, 640, 600, True, True),
                                                 (720, 1280, 600, False, False), (500, 375, 600, True, True),
                                                 (33, 47, 20, False, True)])
def test_image_prep_vs_oracle(ops, H, W, target, flip, rgb):
    """the device front-end (uint8 in, mean-subtracted resized BGR NHWC4 blob out) is bit-equal to
    the numpy restatement of minibatch.py:60-90 + blob.py:35-52."""
    from oracle import data as odata
    rng = np.random.default_rng(H + W)
    u8 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    means = np.array([102.9801, 115.9465, 122.7717], np.float32)
    src = u8 if rgb else np.ascontiguousarray(u8[:, :, ::-1])          # the oracle takes RGB file order
    want, scale = odata.minibatch_image(u8, means, target, flipped=flip)
    blob, (ho, wo, sc) = ops.image_prep(torch.from_numpy(src).to(DEV), means, target, flipped=flip, rgb=rgb)
    assert (ho, wo) == want.shape[:2] and abs(sc - scale) < 1e-6
    got = blob[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(got[:, :, :3], want)
    assert not got[:, :, 3].any()
    # into a larger zero-padded batch blob (blob.py:19-33)
    big = torch.zeros((1, 4, ho + 5, wo + 9), device=DEV).contiguous(memory_format=torch.channels_last)
    ops.image_prep(torch.from_numpy(src).to(DEV), means, target, flipped=flip, rgb=rgb, blob=big)
    b = big[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(b[:ho, :wo, :3], want) and not b[ho:].any() and not b[:, wo:].any()


def test_get_minibatch_device_equals_host(ops):
    """roi_data_layer: the device front-end produces the blob of the host path (same im_info, gt_boxes, pixels)."""
    import numpy.random as npr
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.roi_data_layer import minibatch as mb
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg.TRAIN.USE_FLIPPED = True
    _, roidb, _, _ = combined_roidb("synthetic_8", training=True)
    for e in (roidb[0], roidb[-1]):                       # the last half of the list are the flipped copies
        npr.seed(0)
        host = mb.get_minibatch([e], 16)
        npr.seed(0)
        dev = mb.get_minibatch_device([e], 16, DEV)
        assert np.array_equal(host["im_info"], dev["im_info"]) and np.array_equal(host["gt_boxes"], dev["gt_boxes"])
        d = dev["data"][0].permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(d[:, :, :3], host["data"][0]) and not d[:, :, 3].any()


def test_l2norm_rows_and_bce_rows_vs_torch(ops):
    """The two fused tail ops == F.normalize / BCEWithLogits expressions of resnet_SGG_emb.py:210-213 and
    faster_rcnn_SGG_emb.py:269 (forward and autograd backward)."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((64, 300), dtype=np.float32)
    x[3] = 0.0                                              # a zero row: the eps clamp
    g = rng.standard_normal((64, 300), dtype=np.float32)
    xt = torch.from_numpy(x).requires_grad_()
    F.normalize(xt, p=2, dim=1).backward(torch.from_numpy(g))
    xd = torch.from_numpy(x).to(DEV).requires_grad_()
    y = ops.l2norm_rows(xd)
    y.backward(torch.from_numpy(g).to(DEV))
    np.testing.assert_allclose(y.detach().cpu().numpy(), F.normalize(torch.from_numpy(x), p=2, dim=1).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(xd.grad[np.arange(64) != 3].cpu().numpy(), xt.grad[np.arange(64) != 3].numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(xd.grad[3].cpu().numpy(), xt.grad[3].numpy(), rtol=1e-5)          # g / eps
    z = (rng.standard_normal((42, 62)) * 3).astype(np.float32)
    t = (rng.uniform(size=(42, 62)) < 0.1).astype(np.float32)
    w = rng.uniform(0.01, 0.05, 42).astype(np.float32)
    zt = torch.from_numpy(z).requires_grad_()
    ref = (F.binary_cross_entropy_with_logits(zt, torch.from_numpy(t), reduction="none").mean(1) * torch.from_numpy(w)).sum()
    (ref * 0.5).backward()
    zd = torch.from_numpy(z).to(DEV).requires_grad_()
    loss = ops.bce_rows(zd, torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    (loss * 0.5).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
    np.testing.assert_allclose(zd.grad.cpu().numpy(), zt.grad.numpy(), rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("B,C,N,H,W", [(2, 256, 256, 38, 63), (1, 128, 128, 75, 125), (3, 64, 128, 7, 7), (1, 32, 16, 5, 6),
                                        (2, 16, 32, 1, 1)])
def test_winograd_3x3_vs_torch_fp32(ops, B, C, N, H, W):
    """Winograd F(2x2,3x3) forward (frozen-filter 3x3 stride-1 pad-1 convolutions) == torch fp32 conv2d + frozen-BN
    scale/shift + ReLU, on even / odd / tiny spatial sizes (the last tile row / column is partial when H or W is odd)."""
    rng = np.random.default_rng(B * 1000 + C + H)
    x = rng.standard_normal((B, C, H, W), dtype=np.float32)
    w = (rng.standard_normal((N, C, 3, 3), dtype=np.float32) / np.sqrt(9 * C)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, N).astype(np.float32)
    sh = rng.uniform(-0.5, 0.5, N).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1)
    xd = torch.from_numpy(x).to(DEV)
    U = ops.winograd_filter(torch.from_numpy(w).to(DEV))
    assert U.shape == (16, N, C)
    y = ops.conv3x3_winograd(xd, U)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    y = ops.conv3x3_winograd(xd, U, torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV), relu=True)
    r2 = F.relu(ref * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(sh).view(1, -1, 1, 1))
    np.testing.assert_allclose(y.cpu().numpy(), r2.numpy(), rtol=2e-5, atol=2e-5)
    # and the same numbers as the direct implicit-GEMM path, to rounding
    d = ops.conv2d(xd, torch.from_numpy(w).to(DEV), torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV), None, 1, 1, relu=True)
    np.testing.assert_allclose(y.cpu().numpy(), d.cpu().numpy(), rtol=2e-5, atol=2e-5)
    # F(4x4,3x3): 4x fewer MACs, transform constants up to 8 and 1/24 -> ~1e-5 of the output scale
    U4 = ops.winograd_filter(torch.from_numpy(w).to(DEV), 4)
    assert U4.shape == (36, N, C)
    y4 = ops.conv3x3_winograd(xd, U4, torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV), relu=True)
    assert np.abs(y4.cpu().numpy() - r2.numpy()).max() <= 1e-4 * max(np.abs(r2.numpy()).max(), 1.0)


@pytest.mark.parametrize("M,K,N,res", [(4788, 1024, 256, False),     # layer3 conv1, two frames: split-K finished in the kernel
                                       (4788, 256, 1024, True),      # layer3 conv3 + residual + ReLU
                                       (18750, 128, 512, True),      # layer2 conv3: 4 stages
                                       (9000, 64, 256, True),        # layer1 conv3 shape class: 2 stages
                                       (2394, 512, 256, False),      # one frame
                                       (62, 300, 1024, False),       # K and M not multiples of the tile / stage
                                       (9000, 300, 36, False)])      # N % 16 != 0, K % 32 != 0, two splits finished in the kernel
def test_pointwise_gemm_kernel_equals_generic_kernel(ops, M, K, N, res):
    """conv_gemm_f32 (lean pointwise / plain-GEMM specialisation: swapped MFMA operands, register epilogue) computes
    bit for bit what conv_igemm_f32 computes -- same K order per accumulator, same split order -- and both match torch
    within fp32 rounding."""
    from i2vsgg_amd._lib import lib
    rng = np.random.default_rng(M + K + N)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)).to(DEV)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(DEV)
    sh = torch.from_numpy(rng.uniform(-0.5, 0.5, N).astype(np.float32)).to(DEV)
    r = torch.from_numpy(rng.standard_normal((M, N), dtype=np.float32)).to(DEV) if res else None
    x4, w4 = x.view(M, K, 1, 1), w.view(N, K, 1, 1)
    r4 = r.view(M, N, 1, 1) if res else None
    out = {}
    from i2vsgg_amd._lib import TUNE
    dma0, atomics0 = lib.i2v_get_tuning(TUNE["I2V_GEMM_DMA"]), lib.i2v_get_tuning(TUNE["I2V_SPLIT_ATOMICS"])
    try:
        assert lib.i2v_get_tuning(20) == 0       # I2V_KGROUPS off (the default): the split across workgroups shares the generic kernel's K cuts
        for mode in (1, 0):
            assert lib.i2v_set_tuning(10, mode) == 0
            out[mode] = ops.conv2d(x4, w4, sc, sh, r4, 1, 0, relu=True).view(M, N).clone()
        # round 6: the three staging forms of conv_gemm_f32 (through registers / LDS-DMA with 32-k / 16-k stages) for the
        # cost model's tile and for every forced tile: the same bits
        assert lib.i2v_set_tuning(10, 1) == 0
        for stg in (0, 1, 2):
            assert lib.i2v_set_tuning(TUNE["I2V_GEMM_DMA"], stg) == 0
            assert torch.equal(ops.conv2d(x4, w4, sc, sh, r4, 1, 0, relu=True).view(M, N), out[0]), ("staging form", stg)
        # a forced tile has its own split-K plan (other K cuts, possibly more than four parts: ordered finish, not atomics):
        # the forms are compared tile by tile
        assert lib.i2v_set_tuning(TUNE["I2V_SPLIT_ATOMICS"], 0) == 0
        for tile in range(6):
            assert lib.i2v_conv_set_tile(tile) == 0
            for stg in (0, 1, 2):
                assert lib.i2v_set_tuning(TUNE["I2V_GEMM_DMA"], stg) == 0
                got = ops.conv2d(x4, w4, sc, sh, r4, 1, 0, relu=True).view(M, N)
                if stg == 0:
                    first = got.clone()
                    np.testing.assert_allclose(first.cpu().numpy(), out[0].cpu().numpy(), rtol=2e-5, atol=2e-5)
                else:
                    assert torch.equal(got, first), ("staging form", stg, "tile", tile)
    finally:
        lib.i2v_set_tuning(10, 1)
        lib.i2v_set_tuning(TUNE["I2V_GEMM_DMA"], dma0)
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_ATOMICS"], atomics0)
        lib.i2v_conv_set_tile(-1)
    assert torch.equal(out[1], out[0])
    ref = x.double() @ w.double().t() * sc.double() + sh.double()
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref)
    np.testing.assert_allclose(out[1].cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,K,N,res,mask", [(4788, 1024, 256, True, False), (2394, 1024, 256, False, False), (4788, 1024, 256, True, True),
                                              (1568, 1024, 512, True, False), (4788, 768, 200, False, False), (6000, 512, 128, True, False)])
def test_intra_workgroup_k_split_equals_memory_split(ops, M, K, N, res, mask):
    """Round-3 review item 5 (I2V_KGROUPS=1; off by default, profiles/r04_kgroups.txt): conv_gemm_f32<.., KG = 4> -- a pointwise GEMM the plan would split over K runs as one 16-wave
    workgroup per tile whose four wave groups take a quarter of K each and meet in LDS (layer3 conv1 of a frame pair: K 1024,
    N 256; its data-gradient twin with the ReLU mask epilogue) -- against (a) the memory-side split it replaces (I2V_KGROUPS=0:
    partial tiles through the workspace, summed in split order), (b) the generic kernel conv_igemm_f32 (I2V_CONV_GEMM=0) and
    (c) float64.  K order: group g sums k in [g K/4, (g+1) K/4) in ascending k, the four partial sums are added
    ((p0 + p1) + p2) + p3; the split forms cut K elsewhere (3 x 352 / 352 / 320 at K = 1024), so results agree to summation-order
    rounding (1e-6 of the output scale), not bit for bit.  Deterministic: two launches give the same bits."""
    from i2vsgg_amd._lib import TUNE, lib
    rng = np.random.default_rng(M + K + N)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)).to(DEV)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(DEV)
    sh = torch.from_numpy(rng.uniform(-0.5, 0.5, N).astype(np.float32)).to(DEV)
    r4 = torch.from_numpy(rng.standard_normal((M, N), dtype=np.float32)).to(DEV).view(M, N, 1, 1) if res else None
    mk = torch.from_numpy((rng.uniform(size=(M, N)) < 0.7).astype(np.float32)).to(DEV).view(M, N, 1, 1) if mask else None

    def run():
        if mask:        # the data-gradient epilogue: gx = mask > 0 ? (g W) * s + res : 0  (i2v_conv_dgrad_fused, pointwise)
            wt = w.t().contiguous().view(K, N, 1, 1)            # dgrad of a (K <- N) filter is a GEMM with the filter transposed
            return ops._dgrad_fused(x.view(M, K, 1, 1), wt, (M, N, 1, 1), 0, out_scale=sc, res=r4, mask=mk).reshape(M, N).clone()
        return ops.conv2d(x.view(M, K, 1, 1), w.view(N, K, 1, 1), sc, sh, r4, 1, 0, relu=True).view(M, N).clone()

    out = {}
    try:
        for name, key, val in (("groups", "I2V_KGROUPS", 1), ("memory", "I2V_KGROUPS", 0), ("generic", "I2V_CONV_GEMM", 0)):
            old = lib.i2v_get_tuning(TUNE[key])
            assert lib.i2v_set_tuning(TUNE[key], val) == 0
            try:
                out[name] = run()
                if name == "groups":
                    assert torch.equal(run(), out[name])        # no atomics, fixed order
            finally:
                lib.i2v_set_tuning(TUNE[key], old)
    finally:
        pass
    ref = x.double() @ w.double().t()
    if mask:
        ref = ref * sc.double() + (r4.view(M, N).double() if res else 0)
        ref = torch.where(mk.view(M, N) > 0, ref, torch.zeros_like(ref))
    else:
        ref = ref * sc.double() + sh.double()
        if res:
            ref = ref + r4.view(M, N).double()
        ref = torch.relu(ref)
    scale = float(ref.abs().max())
    for name in ("memory", "generic"):
        assert float((out["groups"] - out[name]).abs().max()) <= 1e-6 * scale, name
    assert float((out["groups"].double() - ref).abs().max()) <= 2e-6 * scale
    # the launch plan: no workspace, no clear -- the partial tiles never leave the CU.  (Tiles per shape: 240 of 80x64, 200 of
    # 48x64, 240 of 80x64 with the mask epilogue, 200 of 64x64, then two shapes the form does not take -- K = 768 is no multiple
    # of 128; 6000 x 128 needs > 256 tiles of the smallest tile that would fill the chip -- which keep the split across workgroups.)
    if K % 128 == 0 and not mask and M < 6000:
        assert lib.i2v_get_tuning(TUNE["I2V_KGROUPS"]) == 0 and lib.i2v_conv_split_workspace_bytes(1, 1, M, K, N, 1, 1, 1, 0) > 0
        lib.i2v_set_tuning(TUNE["I2V_KGROUPS"], 1)
        try:
            assert lib.i2v_conv_split_workspace_bytes(1, 1, M, K, N, 1, 1, 1, 0) == 0
        finally:
            lib.i2v_set_tuning(TUNE["I2V_KGROUPS"], 0)


def test_pointwise_gemm_kernel_batched_planes(ops):
    """The 36 element-wise planes of a Winograd F(4x4,3x3) layer3 convolution as one batched launch on conv_gemm_f32."""
    from i2vsgg_amd._lib import lib, ptr, stream
    rng = np.random.default_rng(11)
    nb, M, K, N = 36, 320, 256, 256
    a = torch.from_numpy(rng.standard_normal((nb, M, K), dtype=np.float32)).to(DEV)
    b = torch.from_numpy((rng.standard_normal((nb, N, K), dtype=np.float32) / 16).astype(np.float32)).to(DEV)
    out = {}
    try:
        for mode in (1, 0):
            lib.i2v_set_tuning(10, mode)
            c = torch.empty((nb, M, N), device=DEV)
            assert lib.i2v_gemm_nt_batched(ptr(a), ptr(b), ptr(c), M, N, K, nb, M * K, N * K, M * N, None, 0, stream()) == 0
            out[mode] = c
    finally:
        lib.i2v_set_tuning(10, 1)
    assert torch.equal(out[1], out[0])
    ref = torch.bmm(a.double(), b.double().transpose(1, 2)).float()
    np.testing.assert_allclose(out[1].cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("n_img,P", [(2, 75), (1, 608), (2, 9375)])
def test_dstyle_fused_equals_projections_plus_pooling(ops, n_img, P):
    """netD_style's fused kernel (two projections + product + rank / spatial sums, resnet_instance_styleD_bilinear.py:
    122-131) against the two-GEMM + pooling-pass form it replaces: z within fp32 summation-order noise, every gradient
    (input rows, both filters, both biases) likewise; a forward-only call gives the same z without writing projections."""
    rng = np.random.default_rng(n_img * 1000 + P)
    dim, rank, K = 512, 5, 512
    M, N = n_img * P, dim * rank
    mk = lambda shape, s: torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float32)).to(DEV)
    rows0, w1, b1, w2, b2 = mk((M, K), 1.0), mk((N, K), 0.03), mk((N,), 0.02), mk((N, K), 0.03), mk((N,), 0.02)
    gz = mk((n_img, dim), 1.0)
    out = {}
    for fused in (True, False):
        t = [a.clone().requires_grad_() for a in (rows0, w1, b1, w2, b2)]
        if fused:
            z = ops.dstyle_fused(t[0], t[1], t[2], t[3], t[4], n_img, dim, rank)
        else:
            x1, x2 = ops.linear(t[0], t[1], t[2]), ops.linear(t[0], t[3], t[4])
            z = ops.dstyle_pool(x1.view(n_img, P, N), x2.view(n_img, P, N), dim, rank)
        z.backward(gz)
        out[fused] = [z.detach()] + [a.grad for a in t]
    for a, b in zip(out[True], out[False]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-6, (float((a - b).abs().max()), scale)
    with torch.no_grad():
        z0 = ops.dstyle_fused(rows0, w1, b1, w2, b2, n_img, dim, rank)
    assert torch.equal(z0, out[True][0])
    ref = (((rows0.double() @ w1.double().t() + b1.double()) * (rows0.double() @ w2.double().t() + b2.double()))
           .view(n_img, P, dim, rank).sum((1, 3)))
    assert float((z0.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def _layer_reference(layer, x):
    """Plain torch of a make_layer() stack (frozen BN as scale / shift), in float64: MIOpen's fp32 backward-data picks
    its solver by what the process ran before, and some of them are good to 1e-2 only."""
    for blk in layer:
        def cbr(h, conv, bn, relu, stride=1, pad=0):
            s, b = bn.folded()
            h = F.conv2d(h, conv.weight.double(), None, stride, pad) * s.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
            return F.relu(h) if relu else h
        h = cbr(x, blk.conv1, blk.bn1, True, blk.stride)
        h = cbr(h, blk.conv2, blk.bn2, True, 1, 1)
        h = cbr(h, blk.conv3, blk.bn3, False)
        skip = x if blk.downsample is None else cbr(x, blk.downsample[0], blk.downsample[1], False, blk.stride)
        x = F.relu(h + skip)
    return x


@pytest.mark.parametrize("cin,planes,hw,stride", [(64, 64, (30, 44), 1), (256, 64, (22, 36), 1), (512, 128, (14, 20), 1),
                                                  (256, 128, (28, 40), 2), (512, 256, (15, 25), 2)])
def test_trained_bottleneck_stack_as_fused_autograd_nodes(cin, planes, hw, stride):
    """ops._BottleneckFn (BN scale / ReLU mask / skip gradient folded into the data- and filter-gradient kernels, block-to-
    block hand-over of pre-masked gradients) against (a) plain torch fp32 autograd and (b) the layer-by-layer form
    (I2V_BLOCK_FUSED=0) on a 3-block layer: output, input gradient and every filter gradient."""
    from i2vsgg_amd import ops
    from i2vsgg_amd.model.faster_rcnn.layers import make_layer
    torch.manual_seed(3)
    layer, _ = make_layer(cin, planes, 3, stride)
    layer = layer.to(DEV)
    for m in layer.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 2.0); m.running_mean.normal_(0, 0.2)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.invalidate()
    x0 = torch.relu(torch.randn(2, cin, *hw, device=DEV)).contiguous(memory_format=torch.channels_last)
    ohw = tuple((d - 1) // stride + 1 for d in hw)          # 1x1 / stride s / pad 0
    gout = torch.randn(2, planes * 4, *ohw, device=DEV).contiguous(memory_format=torch.channels_last)
    params = [p for p in layer.parameters() if p.requires_grad]

    def run(fn):
        x = x0.clone().requires_grad_(True)
        for p in params:
            p.grad = None
        y = fn(x)
        (y * gout).sum().backward()
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params]

    assert ops.BLOCK_FUSED and all(b._fused() for b in layer)
    assert layer[0]._next[0] is layer[1] and layer[1].in_relu and not layer[0].in_relu
    fused = run(layer)
    ops.BLOCK_FUSED = False
    try:
        plain = run(layer)
    finally:
        ops.BLOCK_FUSED = True
    xd = x0.double().requires_grad_(True)
    for p in params:
        p.grad = None
    yd = _layer_reference(layer, xd)
    (yd * gout.double()).sum().backward()
    ref = [yd.detach().float(), xd.grad.float()] + [p.grad.clone() for p in params]
    names = ["out", "gx"] + [n for n, p in layer.named_parameters() if p.requires_grad]

    def close(a, b, tol):
        """A pre-activation within rounding of zero takes the other side of its ReLU in one of the two computations (the
        filter gradients are summed with atomics: even the same path does not repeat bit for bit); the gradient behind that
        pixel then differs by O(1) and the difference spreads, attenuated, through the 3x3 and 1x1 layers below it.  So:
        small in the L2 sense (1 in 8 runs reaches 1e-2 on the 14x20 case; a missing mask / scale / skip term is >= 0.3 there), and no more than 0.1 % of the elements
        off by 5 % of the tensor's range."""
        scale = float(b.abs().max()) + 1e-12
        err = (a - b).abs()
        frac = float((err > 5e-2 * scale).float().mean())
        l2 = float(err.norm() / (b.norm() + 1e-12))
        return frac <= 1e-3 and l2 <= tol, (frac, l2)

    for n, a, b, c in zip(names, fused, plain, ref):
        ok, why = close(a, b, 5e-2)
        assert ok, ("fused vs layer-by-layer", n, why)
        ok, why = close(a, c, 5e-2)
        assert ok, ("fused vs torch float64", n, why)
    # the forward has no knife edge that matters at this tolerance
    assert float((fused[0] - ref[0]).abs().max()) <= 1e-3 * float(ref[0].abs().max())


def test_conv_dgrad_fused_and_wgrad_scaled_vs_torch():
    """The C-ABI pieces alone: gx = mask>0 ? (dgrad(gy*gs, w)*os + res) : 0 and gw = rs * wgrad(x, gy), 1x1 and 3x3."""
    from i2vsgg_amd import ops
    torch.manual_seed(5)
    for (cin, cout, k, pad, hw) in [(64, 256, 1, 0, (25, 31)), (256, 64, 1, 0, (25, 31)), (64, 64, 3, 1, (17, 23)), (1024, 256, 1, 0, (38, 63))]:
        B = 2
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)
        x = cl(torch.randn(B, cin, *hw, device=DEV))
        w = cl(torch.randn(cout, cin, k, k, device=DEV) * 0.05)
        gy = cl(torch.randn(B, cout, *hw, device=DEV))
        gs, osc = torch.rand(cout, device=DEV) + 0.5, torch.rand(cin, device=DEV) + 0.5
        res, mask = cl(torch.randn(B, cin, *hw, device=DEV)), cl(torch.randn(B, cin, *hw, device=DEV))
        got = ops._dgrad_fused(gy, w, x.shape, pad, gs, osc, res, mask)
        want = F.conv_transpose2d(gy * gs.view(1, -1, 1, 1), w, None, 1, pad) * osc.view(1, -1, 1, 1) + res
        want = torch.where(mask > 0, want, torch.zeros_like(want))
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()), (cin, cout, k)
        plain = ops._dgrad_fused(gy, w, x.shape, pad)
        assert float((plain - F.conv_transpose2d(gy, w, None, 1, pad)).abs().max()) <= 1e-4 * float(plain.abs().max())
        gw = ops._wgrad_scaled(x, gy, w.shape, pad, gs)
        wref = torch.nn.grad.conv2d_weight(x, w.shape, gy, 1, pad) * gs.view(-1, 1, 1, 1)
        assert float((gw - wref).abs().max()) <= 1e-4 * float(wref.abs().max()), (cin, cout, k)


@pytest.mark.parametrize("B,C,N,H,W", [(2, 64, 64, 30, 44), (1, 128, 64, 13, 21), (2, 256, 256, 38, 63), (4, 64, 128, 75, 125)])
def test_winograd_filter_gradient_vs_torch(B, C, N, H, W):
    """i2v_conv3x3_winograd4_wgrad (36 plane GEMMs over the 4x4 tiles + the two operand transforms + the 36 -> 9 transform)
    against float64 torch and against the direct filter-gradient kernel; beta = 1 accumulates; row_scale scales rows."""
    from i2vsgg_amd import ops
    torch.manual_seed(B + C)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x, g = cl(torch.randn(B, C, H, W, device=DEV)), cl(torch.randn(B, N, H, W, device=DEV))
    rs = torch.rand(N, device=DEV) + 0.5
    want = torch.nn.grad.conv2d_weight(x.double(), (N, C, 3, 3), g.double(), 1, 1)
    scale = float(want.abs().max())
    got = ops._conv_wgrad_raw(x, g, (N, C, 3, 3), 1, 1, winograd=True)
    direct = ops._conv_wgrad_raw(x, g, (N, C, 3, 3), 1, 1)
    assert float((direct.double() - want).abs().max()) <= 1e-4 * scale
    assert float((got.double() - want).abs().max()) <= 2e-4 * scale, float((got.double() - want).abs().max()) / scale
    got = ops._conv_wgrad_raw(x, g, (N, C, 3, 3), 1, 1, winograd=True, row_scale=rs)
    assert float((got.double() - want * rs.double().view(-1, 1, 1, 1)).abs().max()) <= 2e-4 * 1.5 * scale


@pytest.mark.parametrize("B,C,N,H,W,k", [(4, 1024, 256, 38, 63, 1), (4, 256, 1024, 38, 63, 1), (2, 128, 512, 75, 125, 1), (4, 64, 64, 75, 125, 3),
                                         (4, 64, 256, 150, 250, 1),     # layer1's expansion at 4 frames: 4 tiles x 254 splits
                                         (3, 512, 128, 75, 125, 1),     # 16 tiles x 55 splits: an odd group count
                                         (1, 256, 256, 38, 63, 3),      # Winograd planes: 36 x 16 tiles x 1 split; 18 splits direct
                                         (1, 192, 64, 20, 33, 1)])      # 3 tiles x 5 splits: fewer groups than XCDs, W % 8 != 0
def test_filter_gradient_split_groups_on_one_xcd(B, C, N, H, W, k):
    """Filter gradients take the dispatch-index remap (a split's tiles share an XCD): same result as with the remap off, and
    both equal float64 torch.  Round 2's remap needed a group count (splits x planes) that is a multiple of 8; round 4 deals
    every XCD a contiguous range of the group-major order, for ANY group count (odd, fewer than 8, a launch padded by up to 7
    workgroups that leave at once)."""
    from i2vsgg_amd import _lib, ops
    torch.manual_seed(7)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x, g = cl(torch.randn(B, C, H, W, device=DEV)), cl(torch.randn(B, N, H, W, device=DEV))
    want = torch.nn.grad.conv2d_weight(x.double(), (N, C, k, k), g.double(), 1, k // 2)
    scale = float(want.abs().max())
    assert _lib.lib.i2v_get_tuning(14) == 1
    on = ops._conv_wgrad_raw(x, g, (N, C, k, k), 1, k // 2)
    _lib.lib.i2v_set_tuning(14, 0)
    try:
        off = ops._conv_wgrad_raw(x, g, (N, C, k, k), 1, k // 2)
    finally:
        _lib.lib.i2v_set_tuning(14, 1)
    assert float((on.double() - want).abs().max()) <= 1e-4 * scale
    assert float((off.double() - want).abs().max()) <= 1e-4 * scale
    assert float((on - off).abs().max()) <= 2e-5 * scale          # atomics: summation order differs


@pytest.mark.parametrize("B,C,N,H,W", [(2, 256, 128, 38, 63), (1, 72, 196, 19, 23), (3, 64, 64, 7, 7), (1, 1024, 256, 5, 3)])
def test_filter_gradient_lds_dma_staging_is_bit_equal(B, C, N, H, W):
    """Round 6: the pointwise filter gradient stages its tiles by LDS-DMA (I2V_TUNE_WGRAD_DMA, conv_wgrad2_f32<.., DMA>): a
    [pixel][column] LDS image, the group swizzle on the source column, the stage offset in the request's scalar offset.  Same MFMA
    order per accumulator as the register-staged kernel: the SAME BITS with ordered sums, for every tile shape, on channel counts
    that leave partial tiles (72, 196), pixel counts that leave a partial last stage, and the 36-plane batched form the Winograd
    filter gradient uses; both within 1e-5 of float64 torch."""
    from i2vsgg_amd import ops
    from i2vsgg_amd._lib import TUNE, lib
    torch.manual_seed(11)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x, g = cl(torch.randn(B, C, H, W, device=DEV)), cl(torch.randn(B, N, H, W, device=DEV))
    want = torch.nn.grad.conv2d_weight(x.double(), (N, C, 1, 1), g.double(), 1, 0)
    scale = float(want.abs().max())
    ctx = ops.LaunchContext(DEV, ordered=True)
    keys = [TUNE["I2V_WGRAD_DMA"], TUNE["I2V_WGRAD_V2"]]
    old = [lib.i2v_get_tuning(k) for k in keys]
    assert old[0] == 1                                      # the default
    try:
        for tiles in (1, 2, 3):                             # 64x64, 128x128, 128x64
            got = {}
            for dma in (0, 1):
                assert lib.i2v_set_tuning(keys[0], dma) == 0 and lib.i2v_set_tuning(keys[1], tiles) == 0
                with ctx, torch.no_grad():
                    got[dma] = (ops._conv_wgrad_raw(x, g, (N, C, 1, 1), 1, 0).clone(),
                                ops._conv_wgrad_raw(x, g, (N, C, 3, 3), 1, 1, winograd=True).clone()
                                if ops._winograd_wgrad_ok(x, (N, C, 3, 3), 1, 1) else None)
            assert torch.equal(got[0][0], got[1][0]), (tiles, float((got[0][0] - got[1][0]).abs().max()))
            assert float((got[1][0].double() - want).abs().max()) <= 1e-5 * scale
            if got[0][1] is not None:
                assert torch.equal(got[0][1], got[1][1]), tiles
    finally:
        for k, v in zip(keys, old):
            lib.i2v_set_tuning(k, v)


def test_pair_gather_and_single_workgroup_bce_match_torch():
    """ops.pair_gather ([obj[ixs] | obj[ixo]] per relation pair, resnet_SGG_emb.py:170-176) forward and backward against
    index_select / cat under autograd (repeated and unused rows, a pad index pair (0, 0)); ops.bce_rows against
    F.binary_cross_entropy_with_logits with the per-frame means folded into row weights, bit-stable across runs."""
    from i2vsgg_amd import ops
    g = torch.Generator().manual_seed(4)
    obj = torch.randn(37, 300, generator=g).to(DEV).requires_grad_()
    ixs = torch.tensor([3, 3, 0, 36, 5, 0, 0, 12, 3], device=DEV)
    ixo = torch.tensor([4, 9, 0, 1, 5, 0, 0, 3, 36], device=DEV)
    gy = torch.randn(9, 600, generator=g).to(DEV)
    out = ops.pair_gather(obj, ixs, ixo)
    out.backward(gy)
    got_g = obj.grad.clone()
    obj.grad = None
    ref = torch.cat((obj.index_select(0, ixs), obj.index_select(0, ixo)), 1)
    ref.backward(gy)
    assert torch.equal(out, ref)
    torch.testing.assert_close(got_g, obj.grad, rtol=1e-6, atol=1e-6)
    assert float(got_g[20].abs().max()) == 0.0                   # a box no pair refers to: zeros, written not accumulated
    z = torch.randn(64, 62, generator=g).to(DEV).requires_grad_()
    t = (torch.rand(64, 62, generator=g) < 0.05).float().to(DEV)
    w = torch.cat((torch.full((40,), 1 / 80.0), torch.full((20,), 1 / 40.0), torch.zeros(4))).to(DEV)      # 2 frames + 4 pad rows
    a = ops.bce_rows(z, t, w)
    a.backward()
    ga = z.grad.clone()
    z.grad = None
    per = torch.nn.functional.binary_cross_entropy_with_logits(z, t, reduction="none").mean(1)
    b = 0.5 * (per[:40].mean() + per[40:60].mean())
    b.backward()
    assert abs(float(a) - float(b)) < 1e-6 * abs(float(b))
    torch.testing.assert_close(ga, z.grad, rtol=1e-5, atol=1e-9)
    assert all(float(ops.bce_rows(z.detach(), t, w)) == float(a) for _ in range(5))


def test_fused_loss_and_target_arithmetic_matches_the_torch_expressions():
    """ops.half_mse / smooth_l1 / bbox_transform / signed_sqrt (one kernel per direction each) against the expressions they
    replace: the discriminator terms of trainval_net_instance_styleD_bilinear.py:276-296, net_utils._smooth_l1_loss (:122-136)
    at both call sites of the detector, bbox_transform_batch (+ the normalisation of proposal_target_layer_cascade.py:104-106)
    and the signed square root of netD_style (resnet_instance_styleD_bilinear.py:137) -- values and gradients."""
    from i2vsgg_amd import ops
    g = torch.Generator().manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)
    # half_mse
    for target in (0.0, 1.0):
        d = torch.sigmoid(rnd(128, 1, 7, 7)).requires_grad_()
        a = ops.half_mse(d, target)
        (3.0 * a).backward()
        ga = d.grad.clone(); d.grad = None
        b = 0.5 * torch.mean((target - d) ** 2)
        (3.0 * b).backward()
        torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
        torch.testing.assert_close(ga, d.grad, rtol=1e-5, atol=1e-10)

    def ref_smooth(pred, tgt, inw, outw, sigma, dim):
        s2 = sigma ** 2
        dd = inw * (pred - tgt)
        ad = dd.abs()
        near = (ad < 1.0 / s2).detach().float()
        loss = outw * (dd * dd * (s2 / 2.0) * near + (ad - 0.5 / s2) * (1.0 - near))
        for i in sorted(dim, reverse=True):
            loss = loss.sum(i)
        return loss.mean()
    # smooth L1: the RPN site ((B,N,4) against (B,N,1) weights, sigma 3, dims [1,2]) and the RCNN site ((R,4), sigma 1, dim [1])
    for shape, wshape, sigma, dim in (((4, 21546, 4), (4, 21546, 1), 3.0, [1, 2]), ((128, 4), (128, 4), 1.0, [1])):
        pred = (rnd(*shape) * 0.5).requires_grad_()
        tgt = rnd(*shape) * 0.5
        inw = (torch.rand(*wshape, generator=g) < 0.3).float().to(DEV)
        outw = inw * 0.01 + (torch.rand(*wshape, generator=g) < 0.1).float().to(DEV) * 0.02
        a = ops.smooth_l1(pred, tgt, inw, outw, sigma)
        a.backward()
        ga = pred.grad.clone(); pred.grad = None
        b = ref_smooth(pred, tgt, inw, outw, sigma, dim)
        b.backward()
        torch.testing.assert_close(a, b, rtol=2e-6, atol=0)
        torch.testing.assert_close(ga, pred.grad, rtol=1e-5, atol=1e-10)
    # bbox transform: shared anchors, batched rois with 5-wide gt rows and normalisation
    from i2vsgg_amd.model.rpn.bbox_transform import _whc
    def ref_bt(ex, gt):
        ew, eh, ecx, ecy = _whc(ex); gw, gh, gcx, gcy = _whc(gt)
        return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), -1)
    def boxes(*sh):
        xy = torch.rand(*sh, 2, generator=g) * 500
        wh = torch.rand(*sh, 2, generator=g) * 300 + 4
        return torch.cat((xy, xy + wh), -1).to(DEV)
    anc, gt = boxes(8151), boxes(4, 8151)
    torch.testing.assert_close(ops.bbox_transform(anc, gt), ref_bt(anc, gt), rtol=1e-6, atol=1e-6)
    rois5 = torch.cat((torch.zeros(4, 32, 1, device=DEV), boxes(4, 32)), 2)
    gt5 = torch.cat((boxes(4, 32), torch.ones(4, 32, 1, device=DEV)), 2)
    means, stds = (0.0, 0.0, 0.0, 0.0), (0.1, 0.1, 0.2, 0.2)
    want = (ref_bt(rois5[:, :, 1:5], gt5[:, :, :4]) - torch.tensor(means, device=DEV)) / torch.tensor(stds, device=DEV)
    torch.testing.assert_close(ops.bbox_transform(rois5[:, :, 1:5], gt5, means, stds), want, rtol=1e-6, atol=1e-6)
    # signed square root
    z = (rnd(4, 512) * 3).requires_grad_()
    with torch.no_grad():
        z[0, :5] = 0.0
    a = ops.signed_sqrt(z)
    w = rnd(4, 512)
    (a * w).sum().backward()
    ga = z.grad.clone(); z.grad = None
    b = torch.sqrt(torch.relu(z)) - torch.sqrt(torch.relu(-z))
    (b * w).sum().backward()
    torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    live = z.detach() != 0
    torch.testing.assert_close(ga[live], z.grad[live], rtol=1e-5, atol=0)
    assert float(ga[~live].abs().max()) == 0.0               # aten gives NaN there (inf * 0 through sqrt'); the kernel gives 0


RETIRED_KNOBS = {"I2V_CONV_SPEC", "I2V_STAGGER", "I2V_GEMM_PERSIST", "I2V_WGRAD_PRIO", "I2V_GEMM_X3", "I2V_FC_FOLD"}
KNOBS = [("I2V_CONV_SPEC", 1), ("I2V_CONV_SPEC", 2), ("I2V_SPLIT_TARGET", 3), ("I2V_SPLIT_TARGET_SKINNY", 3), ("I2V_SPLIT_BELOW", 0),
         ("I2V_SPLIT_BELOW", 2048), ("I2V_SPLIT_ATOMICS", 1), ("I2V_BIG_FC_TILE", -1), ("I2V_WGRAD_V2", 0), ("I2V_WGRAD_V2", 2),
         ("I2V_WGRAD_V2", 3), ("I2V_WINO_ROWS", -1), ("I2V_WINO_ROWS", 3), ("I2V_ROIPOOL_C128", 0), ("I2V_CONV_GEMM", 0),
         ("I2V_STAGGER", 4), ("I2V_ROIALIGN_COLS", 0), ("I2V_ROIALIGN_COLS", 1), ("I2V_WGRAD_PER_CU", 2), ("I2V_WGRAD_XCD", 0), ("I2V_GEMM_PERSIST", 2),
         ("I2V_WGRAD_PRIO", 2), ("I2V_STREAM_TILE", 0), ("I2V_GEMM_X3", 1), ("I2V_KGROUPS", 1), ("I2V_KGROUPS", 2), ("I2V_GEMM_DMA", 0),
         ("I2V_GEMM_DMA", 2), ("I2V_SPLIT_ATOMICS", 0), ("I2V_WGRAD_ORDERED_GFLOP", 0), ("I2V_WGRAD_DMA", 0),
         ("I2V_ROIALIGN_BWD", 0)]


def test_every_tuning_knob_keeps_the_results(ops):
    """Round-2 review: the I2V_* switches select kernels and launch shapes, and only their defaults ran anywhere.  Every knob at
    a non-default value (i2v_set_tuning, the call the environment variables are forwarded to): pointwise / strided / 3x3
    (Winograd) / 5x5 forward, data and filter gradients, the skinny split-K GEMM of the relation head, ROIPool and RoIAlignAvg give
    what the defaults give -- bit for bit where the knob only moves work around, within fp32 summation-order noise where it changes
    a split or a tile."""
    from i2vsgg_amd._lib import TUNE, lib
    rng = np.random.default_rng(5)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh, dtype=np.float32)).to(DEV)
    cl = lambda a: a.contiguous(memory_format=torch.channels_last)
    x = cl(t(2, 256, 38, 63))
    w1, w3, w5 = cl(t(128, 256, 1, 1) / 16), cl(t(256, 256, 3, 3) / 48), cl(t(96, 256, 5, 5) / 80)
    sc, sh = torch.rand(256, device=DEV) + 0.5, torch.rand(256, device=DEV)
    g1 = cl(t(2, 128, 38, 63))
    xs, ws = t(128, 8192), t(512, 8192) / 90
    rois = torch.from_numpy(np.concatenate([rng.integers(0, 2, (48, 1)).astype(np.float32),
                                            np.sort(rng.uniform(0, 600, (48, 2, 2)).astype(np.float32), 1).reshape(48, 4)[:, [0, 2, 1, 3]]], 1)).to(DEV)
    xc = cl(t(2, 1024, 38, 63))
    x2, w2, r2 = cl(t(2, 64, 75, 125)), cl(t(256, 64, 1, 1) / 8), cl(t(2, 256, 75, 125))
    gra = cl(t(48, 1024, 7, 7))

    def run():
        out = {}
        out["pointwise"] = ops.conv2d(x, w1, sc[:128], sh[:128], None, 1, 0, relu=True)
        out["strided"] = ops.conv2d(x, w1, sc[:128], sh[:128], None, 2, 0, relu=True)
        out["winograd"] = ops.conv2d(x, w3, sc, sh, None, 1, 1, relu=True, winograd=True)
        out["direct3x3"] = ops.conv2d(x, w3, sc, sh, None, 1, 1, relu=True)
        out["conv5x5s2"] = ops.conv2d(x, w5, None, None, None, 2, 2)
        out["dgrad"] = ops._conv_dgrad_raw(g1, w1, tuple(x.shape), 1, 0)
        out["wgrad"] = ops._conv_wgrad_raw(x, g1, (128, 256, 1, 1), 1, 0)
        out["skinny"] = ops.linear(xs, ws)
        out["expand"] = ops.conv2d(x2, w2, sc, sh, r2, 1, 0, relu=True)          # K = 64 over 18750 rows: the HBM-bound tile rule
        out["roi_pool"] = ops.roi_pool(xc, rois, 7, 7, 1.0 / 16, out_nchw=False)
        out["roi_align"] = ops.roi_align(xc, rois, 7, 7, 1.0 / 16)
        xg = xc.detach().clone().requires_grad_()
        ops.roi_align(xg, rois, 7, 7, 1.0 / 16).backward(gra)
        out["roi_align_bwd"] = xg.grad
        return {k: v.clone() for k, v in out.items()}

    base = run()
    assert all(torch.isfinite(v).all() for v in base.values())
    for name, value in KNOBS:
        key = TUNE[name]
        old = lib.i2v_get_tuning(key)
        if name in RETIRED_KNOBS:
            # the experiment kernels left the library (round 6): their reserved keys accept nothing but "off", loudly
            assert lib.i2v_set_tuning(key, value) == -4 and b"experiment" in lib.i2v_last_error()
            assert lib.i2v_get_tuning(key) == old
            continue
        try:
            assert lib.i2v_set_tuning(key, value) == 0
            got = run()
        finally:
            lib.i2v_set_tuning(key, old)
        tol = 2e-4 if name == "I2V_GEMM_X3" else 2e-5
        for k in base:
            err = _rel_err(got[k].cpu().numpy(), base[k].cpu().numpy())
            assert err < tol, (name, value, k, err)
    again = run()
    for k in ("pointwise", "strided", "winograd", "roi_pool", "roi_align", "roi_align_bwd"):      # the launches without split-K atomics
        assert torch.equal(again[k], base[k]), k          # the knobs are back at their defaults


def test_fused_adam_equals_torch_adam():
    """train.FusedAdam (i2v_adam_multi, step count in device memory) follows torch.optim.Adam on the same param groups for five
    steps -- tensors below and above a launch table's worth, with and without weight decay, a parameter that never gets a
    gradient -- and its state_dict loads into torch.optim.Adam (and back) with the trajectories still together."""
    from i2vsgg_amd import train
    torch.manual_seed(0)
    shapes = [(7,), (300, 40), (64, 3, 3, 3), (1100, 1000), (5,)] + [(33,)] * 50
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    named = [("w%d.bias" % i if len(s) == 1 else "w%d.weight" % i, p) for i, (s, p) in enumerate(zip(shapes, ps))]
    opt = train.FusedAdam(named, 3e-3, weight_decay=5e-3)
    ref = torch.optim.Adam([{"params": [q], "lr": it["lr"], "weight_decay": it["wd"]} for q, it in zip(qs, opt.items)])
    assert {it["lr"] for it in opt.items} == {3e-3, 6e-3} or not cfg_double_bias()          # biases at twice the rate (DOUBLE_BIAS)

    def one(k):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if i == 4:                       # never gets a gradient: both optimizers leave it alone
                p.grad = q.grad = None
                continue
            g = torch.randn(p.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(100 * k + i))
            p.grad, q.grad = g.clone(), g.clone()
        opt.step()
        ref.step()

    for k in range(5):
        one(k)
        # the early steps are where a float32 bias correction shows (1 - 0.999f^t is off by ~3e-5 at t = 1; round-3 advice): the
        # scalars of a step are computed in double as torch computes them, so the trajectories agree to rounding from step one
        for p, q in zip(ps, qs):
            assert _rel_err(p.detach().cpu().numpy(), q.detach().cpu().numpy()) < 5e-7, k
    assert int(opt.t.item()) == 5 and torch.equal(ps[4], qs[4])
    # state_dict: torch.optim.Adam's layout, both ways; no state for the parameter that never received a gradient
    sd = opt.state_dict()
    assert 4 not in sd["state"] and set(sd["state"]) == set(ref.state_dict()["state"])
    ref2 = torch.optim.Adam([{"params": [q]} for q in qs])
    ref2.load_state_dict({"state": {i: st for i, st in sd["state"].items() if i != 4},
                          "param_groups": [{k: v for k, v in g.items() if k != "name"} for g in sd["param_groups"]]})
    opt2 = train.FusedAdam(named, 1.0)
    opt2.load_state_dict(ref.state_dict())
    assert int(opt2.t.item()) == 5 and opt2.items[1]["lr"] == opt.items[1]["lr"]
    for k in range(5, 8):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if i == 4:
                continue
            g = torch.randn(p.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(100 * k + i))
            p.grad, q.grad = g.clone(), g.clone()
        opt2.step()
        ref2.step()
    for p, q in zip(ps, qs):
        assert _rel_err(p.detach().cpu().numpy(), q.detach().cpu().numpy()) < 3e-6


def cfg_double_bias():
    from i2vsgg_amd.model.utils.config import cfg
    return bool(cfg.TRAIN.DOUBLE_BIAS)
