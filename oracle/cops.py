"""ctypes binding of oracle/c/oracle_ops.c (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ops.so")
_lib = None


def _digest():
    import hashlib
    h = hashlib.sha256()
    for f in (os.path.join(_HERE, "c", "oracle_ops.c"), os.path.join(_HERE, "Makefile")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False):
    """Compile the C restatement with gcc (a few hundred ms).  The shipped .so is reused when it was built from exactly
    these source bytes (a stamp beside it; mtimes say nothing after a copy to another box) -- a test process that has
    already initialised the GPU must not start a compiler."""
    stamp = _SO + ".sha256"
    want = _digest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if force or not os.path.exists(_SO) or have != want:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
        with open(stamp, "w") as f:
            f.write(want)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_nms_sorted.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def nms_sorted(dets, thresh):
    """nms_cpu.py:6-34 on score-sorted rows; returns int32 keep indices."""
    dets, p = _f(dets)
    n = dets.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int32)
    k = lib().oracle_nms_sorted(p, ctypes.c_int(n), ctypes.c_float(thresh), _i(keep))
    return keep[:k].copy()


def roi_align_fwd(feat, rois, ah, aw, scale):
    """roi_align.c:80-136; feat NCHW, rois (R,5) -> (R,C,ah,aw).  Pinned: bit-equal to the reference's compiled function
    (tests/golden/roi_align_fwd.npz, oracle/build_ref.py)."""
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, C, ah, aw), dtype=np.float32)
    lib().oracle_roi_align_fwd(pf, pr, R, C, H, W, ah, aw, ctypes.c_float(scale),
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def roi_align_bwd(gout, rois, feat_shape, scale):
    """roi_align_kernel.cu:94-143 in serial order.  Pinned as the transpose of the pinned forward (tests/test_oracle_golden.py)."""
    gout, pg = _f(gout)
    rois, pr = _f(rois)
    B, C, H, W = feat_shape
    R, _, ah, aw = gout.shape
    gin = np.zeros(feat_shape, dtype=np.float32)
    lib().oracle_roi_align_bwd(pg, pr, R, C, H, W, ah, aw, ctypes.c_float(scale),
                               gin.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return gin


def avgpool2x2_fwd(x):
    x, px = _f(x)
    lead, (ah, aw) = x.shape[:-2], x.shape[-2:]
    n = int(np.prod(lead)) if lead else 1
    y = np.empty(lead + (ah - 1, aw - 1), dtype=np.float32)
    lib().oracle_avgpool2x2_fwd(px, n, ah, aw, y.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return y


def avgpool2x2_bwd(gy):
    gy, pg = _f(gy)
    lead, (ph, pw) = gy.shape[:-2], gy.shape[-2:]
    n = int(np.prod(lead)) if lead else 1
    gx = np.empty(lead + (ph + 1, pw + 1), dtype=np.float32)
    lib().oracle_avgpool2x2_bwd(pg, n, ph + 1, pw + 1, gx.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return gx


def roi_align_sampled_fwd(feat, rois, ph, pw, scale, sampling_ratio):
    """``model._C.roi_align_forward`` (roi_layers/roi_align.py:20; maskrcnn-benchmark ROIAlign_cuda.cu, source absent
    from the reference tree): feat NCHW, rois (R,5) -> (R,C,ph,pw).  Parity unpinned."""
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), dtype=np.float32)
    lib().oracle_roi_align_sampled_fwd(pf, pr, R, C, H, W, ph, pw, ctypes.c_float(scale), int(sampling_ratio),
                                       out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def roi_align_sampled_bwd(gout, rois, feat_shape, scale, sampling_ratio):
    """``model._C.roi_align_backward`` (roi_layers/roi_align.py:31-42) in serial order.  Parity unpinned."""
    gout, pg = _f(gout)
    rois, pr = _f(rois)
    B, C, H, W = feat_shape
    R, _, ph, pw = gout.shape
    gin = np.zeros(feat_shape, dtype=np.float32)
    lib().oracle_roi_align_sampled_bwd(pg, pr, R, C, H, W, ph, pw, ctypes.c_float(scale), int(sampling_ratio),
                                       gin.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return gin


def roi_align_avg_fwd(feat, rois, ph, pw, scale):
    """RoIAlignAvg (roi_align/modules/roi_align.py:18-29): align to (ph+1,pw+1), 2x2 s1 mean."""
    return avgpool2x2_fwd(roi_align_fwd(feat, rois, ph + 1, pw + 1, scale))


def roi_align_avg_bwd(gout, rois, feat_shape, scale):
    return roi_align_bwd(avgpool2x2_bwd(gout), rois, feat_shape, scale)


def roi_pool_fwd(feat, rois, ph, pw, scale):
    """roi_pooling_kernel.cu:24-93 -> (out, argmax[plane index or -1]).  Parity unpinned."""
    feat, pf = _f(feat)
    rois, pr = _f(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, C, ph, pw), dtype=np.float32)
    arg = np.empty((R, C, ph, pw), dtype=np.int32)
    lib().oracle_roi_pool_fwd(pf, pr, R, C, H, W, ph, pw, ctypes.c_float(scale),
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i(arg))
    return out, arg


def roi_pool_bwd(gout, rois, argmax, feat_shape):
    gout, pg = _f(gout)
    rois, pr = _f(rois)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    B, C, H, W = feat_shape
    R, _, ph, pw = gout.shape
    gin = np.zeros(feat_shape, dtype=np.float32)
    lib().oracle_roi_pool_bwd(pg, pr, _i(argmax), R, C, H, W, ph, pw,
                              gin.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return gin


def decode_clip(anchors, deltas, im_h, im_w):
    """bbox_transform.py:77-103,125-133 for one image: (N,4),(N,4) -> (N,4)."""
    anchors, pa = _f(anchors)
    deltas, pd = _f(deltas)
    n = anchors.shape[0]
    out = np.empty((n, 4), dtype=np.float32)
    lib().oracle_decode_clip(pa, pd, n, ctypes.c_float(im_h), ctypes.c_float(im_w),
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out
