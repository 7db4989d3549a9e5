"""Helpers shared by tests/test_oracle_golden.py (the oracle) and tests/test_gpu_kernels.py (the HIP kernels): both are held
to tests/golden/roi_align_fwd.npz -- outputs of the reference's own ``ROIAlignForwardCpu`` (roi_align.c:80-136, tier
"extracted", oracle/build_ref.py) -- and to the transpose of that pinned forward for the backward."""
import numpy as np


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_roi_align_against_golden(g, C, fwd8, avg7, p7):
    """Shared by this file (oracle) and tests/test_gpu_kernels.py (HIP): every stored form of one golden case."""
    tag = "c%d" % C
    for key, got in (("a8", fwd8), ("avg7", avg7), ("p7", p7)):
        if got is None:
            continue
        name = "%s_%s" % (tag, key)
        assert tuple(got.shape) == tuple(g[name + "_shape"]) and got.dtype == np.float32, name
        assert _sha(got) == str(g[name + "_sha256"]), name
        if name in g.files:
            assert np.array_equal(got, g[name]), name
        else:
            assert np.array_equal(got.reshape(-1)[::211], g[name + "_sample"]), name


def roi_align_matrix(rois, B, H, W, ah, aw, scale, fwd):
    """The forward is linear in the map and acts on every channel alike: probing it with a map whose channel j is the
    j-th unit image gives its whole coefficient matrix F[(r, ph, pw), (y, x)] in one call per frame (rois of other frames
    give zero rows there).  Returned per frame, float64 (the entries are the forward's own fp32-rounded coefficients)."""
    eye = np.eye(H * W, dtype=np.float32).reshape(1, H * W, H, W)
    mats = []
    for b in range(B):
        r = rois[rois[:, 0] == b].copy()
        r[:, 0] = 0
        out = fwd(eye, r, ah, aw, scale)                               # (Rb, H*W, ah, aw)
        mats.append(out.transpose(0, 2, 3, 1).reshape(-1, H * W).astype(np.float64))
    return mats


def roi_align_bwd_from_matrix(mats, gout, rois, B, C, H, W):
    gx = np.zeros((B, C, H * W), np.float64)
    for b in range(B):
        g = gout[rois[:, 0] == b].astype(np.float64)                    # (Rb, C, ah, aw)
        g = g.transpose(1, 0, 2, 3).reshape(C, -1)                      # (C, Rb*ah*aw)
        gx[b] = g @ mats[b]
    return gx.reshape(B, C, H, W)
