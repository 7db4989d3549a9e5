"""Hand-worked known answers for the two oracle pieces that have no runnable reference here (ROIAlign,
ROIPool: "parity unpinned").  Every expected number below is derived by hand from
roi_align.c:91-134 / roi_pooling_kernel.cu:45-92 in the comments, not produced by the code under test."""
import numpy as np

from oracle import cops


def test_roi_align_hand_worked():
    # feature f[h][w] = 10*h + w on a 6x8 map (one channel), scale 1.
    H, W = 6, 8
    feat = (10.0 * np.arange(H)[:, None] + np.arange(W)[None, :]).astype(np.float32)[None, None]
    # ROI x1=1,y1=1,x2=4,y2=3 -> roi_w = 4-1+1 = 4, roi_h = 3-1+1 = 3; aligned 3x3 -> bin = roi/(3-1): bw=2, bh=1.5
    # sample points h = ph*1.5 + 1 in {1, 2.5, 4}; w = pw*2 + 1 in {1, 3, 5}
    # f is bilinear-exact for a linear ramp: value = 10*h + w
    rois = np.array([[0, 1, 1, 4, 3]], np.float32)
    out = cops.roi_align_fwd(feat, rois, 3, 3, 1.0)
    exp = np.array([[10 * h + w for w in (1, 3, 5)] for h in (1, 2.5, 4)], np.float32)
    assert np.array_equal(out[0, 0], exp)
    # RoIAlignAvg(2,2): aligned 3x3 then 2x2 stride-1 mean
    avg = cops.roi_align_avg_fwd(feat, rois, 2, 2, 1.0)
    exp_avg = np.array([[(exp[i, j] + exp[i, j + 1] + exp[i + 1, j] + exp[i + 1, j + 1]) / 4 for j in range(2)]
                        for i in range(2)], np.float32)
    assert np.array_equal(avg[0, 0], exp_avg)
    # full-image ROI: x2 = W-1 -> roi_w = W, bin = W/(AW-1): the last sample column is w = W (>= W) -> 0
    full = cops.roi_align_fwd(feat, np.array([[0, 0, 0, W - 1, H - 1]], np.float32), 3, 3, 1.0)
    assert np.all(full[0, 0, 2, :] == 0) and np.all(full[0, 0, :, 2] == 0)
    assert full[0, 0, 0, 0] == 0.0 and full[0, 0, 1, 1] == 10 * 3.0 + 4.0        # h = 1*6/2 = 3, w = 1*8/2 = 4
    # extrapolation at the last row: h = 5.5 -> hstart = min(5, H-2) = 4, h_ratio = 1.5:
    # f(4)*(1-1.5) + f(5)*1.5 = 40*(-0.5) + 50*1.5 = 55 (+ w)
    edge = cops.roi_align_fwd(feat, np.array([[0, 2, 5.5, 2, 5.5]], np.float32), 2, 2, 1.0)
    assert edge[0, 0, 0, 0] == 55.0 + 2.0
    # backward of a single sample with weights (hr, wr) = (0.5, 0): gradient 1 at sample (h=2.5, w=3)
    g = np.zeros((1, 1, 3, 3), np.float32)
    g[0, 0, 1, 1] = 1.0
    gin = cops.roi_align_bwd(g, rois, (1, 1, H, W), 1.0)
    assert gin[0, 0, 2, 3] == 0.5 and gin[0, 0, 3, 3] == 0.5 and gin.sum() == 1.0


def test_roi_align_sampled_properties():
    """roi_layers.ROIAlign restatement (model._C source absent, parity unpinned): properties derivable by hand from the
    published definition.  (1) bilinear sampling reproduces an affine map exactly and the sample grid of a bin is
    symmetric about the bin centre, so every output equals the map at its bin centre; (2) sampling_ratio 0 uses
    ceil(extent / pooled) samples per side; (3) a ROI more than a pixel outside gives zeros; (4) backward is the
    adjoint of forward."""
    from oracle import cops
    H, W, C = 12, 17, 2
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    feat = np.stack([0.25 * yy + 0.5 * xx + 1.0, 2.0 - 0.125 * yy + 0.75 * xx])[None].astype(np.float32)
    rois = np.array([[0, 16.0, 24.0, 208.0, 152.0]], np.float32)           # map coords x 1..13, y 1.5..9.5
    for sr in (0, 1, 2, 3):
        out = cops.roi_align_sampled_fwd(feat, rois, 4, 3, 1 / 16.0, sr)
        bh, bw = 8.0 / 4, 12.0 / 3
        cy = 1.5 + (np.arange(4) + 0.5) * bh
        cx = 1.0 + (np.arange(3) + 0.5) * bw
        want0 = 0.25 * cy[:, None] + 0.5 * cx[None, :] + 1.0
        want1 = 2.0 - 0.125 * cy[:, None] + 0.75 * cx[None, :]
        np.testing.assert_allclose(out[0, 0], want0, rtol=0, atol=2e-6)
        np.testing.assert_allclose(out[0, 1], want1, rtol=0, atol=2e-6)
    # (2) adaptive grid: a 6-px-tall bin row over a map that is 1 on row 3 only.  Bin 0 of a 1x1 pooling of the box
    # y in [0,6] is sampled at gh = ceil(6/1) = 6 rows y = 0.5, 1.5, ... 5.5: rows 2.5 and 3.5 see 0.5 each -> 1/6.
    f = np.zeros((1, 1, 8, 8), np.float32)
    f[0, 0, 3, :] = 1.0
    r = np.array([[0, 0, 0, 6, 6]], np.float32)
    assert abs(cops.roi_align_sampled_fwd(f, r, 1, 1, 1.0, 0)[0, 0, 0, 0] - 1.0 / 6.0) < 1e-7
    assert abs(cops.roi_align_sampled_fwd(f, r, 1, 1, 1.0, 1)[0, 0, 0, 0] - 1.0) < 1e-7     # one sample at y = 3
    assert abs(cops.roi_align_sampled_fwd(f, r, 1, 1, 1.0, 2)[0, 0, 0, 0] - 0.0) < 1e-7     # samples at y = 1.5, 4.5
    # (3) outside: everything beyond W+1 / below -1
    far = np.array([[0, 400, 400, 500, 500], [0, -300, -300, -40, -40]], np.float32)
    assert np.all(cops.roi_align_sampled_fwd(feat, far, 3, 3, 1 / 16.0, 0) == 0)
    # degenerate extent is clamped to one pixel, not zero: a point ROI returns the interpolated value around it
    pt = np.array([[0, 64, 48, 64, 48]], np.float32)                       # (x,y) = (4,3), 1x1 extent
    out = cops.roi_align_sampled_fwd(feat, pt, 1, 1, 1 / 16.0, 0)
    assert abs(out[0, 0, 0, 0] - (0.25 * 3.5 + 0.5 * 4.5 + 1.0)) < 1e-6
    # (4) adjoint
    rng = np.random.default_rng(0)
    fr = rng.standard_normal((2, 3, H, W), dtype=np.float32)
    rr = np.array([[0, 5, 9, 180, 120], [1, -20, 30, 300, 100], [1, 100, 100, 130, 190]], np.float32)
    g = rng.standard_normal((3, 3, 5, 4), dtype=np.float32)
    lhs = float((cops.roi_align_sampled_fwd(fr, rr, 5, 4, 1 / 16.0, 0).astype(np.float64) * g).sum())
    rhs = float((cops.roi_align_sampled_bwd(g, rr, fr.shape, 1 / 16.0, 0).astype(np.float64) * fr).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_roi_pool_hand_worked():
    # 1 channel 4x6 map, values row-major 0..23; scale 1
    feat = np.arange(24, dtype=np.float32).reshape(1, 1, 4, 6)
    # ROI x1=1,y1=0,x2=4,y2=3 -> 4 wide, 4 high; pooled 2x2 -> bins 2x2 pixels
    rois = np.array([[0, 1, 0, 4, 3]], np.float32)
    out, arg = cops.roi_pool_fwd(feat, rois, 2, 2, 1.0)
    # bin (0,0): rows 0-1, cols 1-2 -> max = f[1][2] = 8 ; bin (0,1): cols 3-4 -> f[1][4] = 10
    # bin (1,0): rows 2-3 -> f[3][2] = 20 ; bin (1,1): f[3][4] = 22
    assert np.array_equal(out[0, 0], np.array([[8, 10], [20, 22]], np.float32))
    assert np.array_equal(arg[0, 0], np.array([[8, 10], [20, 22]], np.int32))      # h*W + w of the winner
    # rounding is half away from zero: x1 = 2.5 -> 3 (roi_pooling_kernel.cu:46)
    out2, _ = cops.roi_pool_fwd(feat, np.array([[0, 2.5, 0, 5, 0]], np.float32), 1, 1, 1.0)
    assert out2[0, 0, 0, 0] == 5.0                                                 # row 0, cols 3..5 -> max 5
    # ROI entirely right of the map: empty bins -> 0 and argmax -1 (:66-70)
    out3, arg3 = cops.roi_pool_fwd(feat, np.array([[0, 40, 40, 50, 50]], np.float32), 2, 2, 1.0)
    assert np.all(out3 == 0) and np.all(arg3 == -1)
    # backward routes each bin's gradient to its argmax
    gin = cops.roi_pool_bwd(np.ones((1, 1, 2, 2), np.float32), rois, arg, feat.shape)
    exp = np.zeros((4, 6), np.float32)
    exp[1, 2] = exp[1, 4] = exp[3, 2] = exp[3, 4] = 1
    assert np.array_equal(gin[0, 0], exp)


def test_detection_postprocess_hand_worked():
    """oracle.rpn.detection_postprocess (test_net_instance_styleD_bilinear.py:151-221) on a case small enough to do by
    hand: zero deltas give x1' = ctr - w/2 = x1 and x2' = ctr + w/2 = x2 + 1 (the legacy +1 width convention of
    bbox_transform.py:80-100), then / scale; two heavily overlapping rois of one class -> the lower
    score is suppressed at NMS 0.3; the image-wide cut keeps scores >= the max_per_image-th largest."""
    from oracle import rpn as orpn
    rois = np.array([[0, 10, 10, 109, 109], [0, 12, 12, 111, 111], [0, 300, 200, 399, 299]], np.float32)
    prob = np.array([[0.1, 0.9, 0.0], [0.2, 0.8, 0.0], [0.3, 0.1, 0.6]], np.float32)
    pred = np.zeros((3, 12), np.float32)
    out = orpn.detection_postprocess(rois, prob, pred, 600, 1000, 2.0, False, (0.1, 0.1, 0.2, 0.2), (0, 0, 0, 0), 0.05, 0.3, 100)
    assert len(out) == 3 and out[0].shape == (0, 5)
    # class 1: rois 0, 1, 2 pass the threshold (0.9, 0.8, 0.1); roi 1 overlaps roi 0 (IoU 0.92) and is suppressed
    np.testing.assert_array_equal(out[1], np.array([[5, 5, 55, 55, 0.9], [150, 100, 200, 150, 0.1]], np.float32))
    # class 2: only roi 2 (0.6)
    np.testing.assert_array_equal(out[2], np.array([[150, 100, 200, 150, 0.6]], np.float32))
    # top-2 over the image: threshold = 2nd largest kept score = 0.6
    out = orpn.detection_postprocess(rois, prob, pred, 600, 1000, 2.0, False, (0.1, 0.1, 0.2, 0.2), (0, 0, 0, 0), 0.05, 0.3, 2)
    assert [len(o) for o in out] == [0, 1, 1] and out[1][0, 4] == np.float32(0.9)


def test_image_front_end_restatement():
    """oracle.data (blob.py:35-52 with cv2's float INTER_LINEAR restated): identity at f = 1, a hand-worked 2x upscale
    (half-pixel centres: output 0 sits at source -0.25 -> clamped to pixel 0; output 1 at 0.25 -> 0.75*p0 + 0.25*p1),
    and agreement with an independent implementation of the same sampling rule (torch bilinear, align_corners=False)."""
    import torch
    import torch.nn.functional as F
    from oracle import data as odata
    rng = np.random.default_rng(0)
    im = rng.uniform(-120, 140, (7, 9, 3)).astype(np.float32)
    assert np.array_equal(odata.resize_linear(im, 1.0), im)
    row = np.array([[[0.0], [8.0]]], np.float32)                       # 1 x 2 x 1
    up = odata.resize_linear(np.repeat(row, 2, 0), 2.0)[0, :, 0]
    np.testing.assert_array_equal(up, np.array([0.0, 2.0, 6.0, 8.0], np.float32))
    for f in (1.6, 600.0 / 375.0, 0.75):
        got = odata.resize_linear(im, f)
        t = torch.from_numpy(im).permute(2, 0, 1)[None]
        ref = F.interpolate(t, size=got.shape[:2], scale_factor=None, mode="bilinear", align_corners=False)
        ref = F.interpolate(t, scale_factor=f, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        assert tuple(ref.shape[2:]) == got.shape[:2] or abs(ref.shape[2] - got.shape[0]) <= 1
        if tuple(ref.shape[2:]) == got.shape[:2]:
            np.testing.assert_allclose(got, ref[0].permute(1, 2, 0).numpy(), atol=2e-4)
    u8 = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    out, scale = odata.minibatch_image(u8, (102.9801, 115.9465, 122.7717), 60)
    assert out.shape == (60, 80, 3) and scale == 2.0
    out_f, _ = odata.minibatch_image(u8, (102.9801, 115.9465, 122.7717), 60, flipped=True)
    np.testing.assert_array_equal(out_f, out[:, ::-1])                   # the sampling grid is symmetric
