"""torch-facing wrappers over the C-ABI of libi2vsgg_hip.so.

Conventions
  * activations are torch tensors with the reference's LOGICAL shape (B,C,H,W) held in
    ``channels_last`` memory format, i.e. physically NHWC -- the layout the kernels want;
    ``as_nhwc`` converts (one copy) when a caller hands over a plain NCHW tensor;
  * conv weights keep the reference's logical shape (Cout,Cin,KH,KW), also channels_last,
    which is exactly the (Cout,KH,KW,Cin) K-major filter layout of the implicit GEMM;
  * every function launches on torch's current stream and never synchronises;
  * no CPU fallback: non-CUDA tensors raise.
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_RELU, EPI_RESIDUAL, EPI_SCALE, EPI_ZEROED, LAYOUT_NCHW, LAYOUT_NHWC, check, lib, ptr,
                   stream)

_CL = torch.channels_last


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.I2VError("i2vsgg_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback"
                                % t.device.type)


def as_nhwc(x):
    """(B,C,H,W) tensor -> same logical tensor, physically NHWC (no copy if already so)."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D (B,C,H,W) tensor")
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous(memory_format=_CL)


def _is_nhwc(x):
    return x.dim() == 4 and x.is_contiguous(memory_format=_CL)


_ws_cache = {}
SCRATCH = None      # dict of the active LaunchContext (below); None -> scratch per launch stream


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (tag, owner).  The owner is the active ``LaunchContext`` when a step object has
    installed one, else the launch stream: two pieces of work that may run concurrently on the device (two streams, two
    HIP graphs replayed side by side) never share a scratch buffer -- every kernel that uses one assumes that the
    launches before it on ITS stream are the only other users."""
    cache = SCRATCH
    if cache is None:
        key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
        cache = _ws_cache.setdefault(key, {})
    buf = cache.get(tag)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        cache[tag] = buf
    return buf


# ----------------------------------------------------------------------------- ROI ops
def _rois_f32(rois):
    if rois.dim() != 2 or rois.size(1) != 5:
        # roi_align.c:28-31 returns 0 for a malformed roi tensor; here it is an error
        raise ValueError("rois must be (K,5) [batch_idx,x1,y1,x2,y2]")
    return rois.contiguous().float()


ROIALIGN_BWD_GATHER = True     # False (tests set it): the atomic scatter of round 1, which still serves NCHW / narrow maps
# kernel name -> launches per op, for the profiling tools (bench.py roi_nms_case, tools/roi_nms_pmc_summary.py)
ROIALIGN_BWD_KERNELS = {"roi_align_bwd_row2_kernel": 1}     # (I2V_TUNE_ROIALIGN_BWD = 0 runs roi_align_bwd_row_kernel, round 5's form)
NMS_KERNELS = {"nms_mask_kernel": 1, "nms_scan_super_kernel": 1}      # (I2V_TUNE_NMS_SCAN = 0: nms_scan_pipelined_kernel, round 5's scan)


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale, avg, out_nchw):
        _need_cuda(feat, rois)
        rois = _rois_f32(rois)
        nhwc = _is_nhwc(feat)
        if not nhwc:
            feat = feat.contiguous()
        B, C, H, W = feat.shape
        R = rois.size(0)
        out = torch.empty((R, C, ph, pw), device=feat.device, dtype=torch.float32,
                          memory_format=torch.contiguous_format if out_nchw else _CL)
        check(lib.i2v_roi_align_fwd(ptr(feat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W, ptr(rois), R, ph, pw,
                                    scale, int(avg), ptr(out), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, stream()),
              "roi_align_fwd")
        ctx.save_for_backward(rois)
        ctx.meta = (feat.shape, nhwc, ph, pw, scale, int(avg), out_nchw)
        return out

    @staticmethod
    def backward(ctx, gout):
        (rois,) = ctx.saved_tensors
        shape, nhwc, ph, pw, scale, avg, out_nchw = ctx.meta
        B, C, H, W = shape
        gout = gout.contiguous() if out_nchw else gout.contiguous(memory_format=_CL)
        if ROIALIGN_BWD_GATHER and nhwc and not out_nchw and C % 128 == 0 and W * 512 + 21504 <= 60 * 1024 and pw + avg <= 8 and rois.size(0) > 0:
            # the gather form: every element of the gradient map written once, in the reference's serial order (deterministic,
            # no atomics, no zero-fill, no workspace: one kernel since round 5)
            gfeat = torch.empty(shape, device=gout.device, dtype=torch.float32, memory_format=_CL)
            check(lib.i2v_roi_align_bwd_gather(ptr(gout), ptr(rois), rois.size(0), ph, pw, scale, avg, ptr(gfeat), B, C, H, W,
                                               None, 0, stream()), "roi_align_bwd_gather")
            return gfeat, None, None, None, None, None, None
        # the scatter accumulates with atomics into zeros: inside a step they come from the step's pre-zeroed arena (one
        # clear per step for every atomically accumulated output) instead of a fill kernel of their own
        gfeat = ARENA.take(B, C, H, W) if (ARENA is not None and nhwc) else None
        if gfeat is None:
            gfeat = torch.empty(shape, device=gout.device, dtype=torch.float32,
                                memory_format=_CL if nhwc else torch.contiguous_format).zero_()
        check(lib.i2v_roi_align_bwd(ptr(gout), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, ptr(rois), rois.size(0), ph, pw,
                                    scale, avg, ptr(gfeat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W, stream()),
              "roi_align_bwd")
        return gfeat, None, None, None, None, None, None


def roi_align(feat, rois, pooled_h, pooled_w, spatial_scale, avg=True, out_nchw=False):
    """Legacy ROIAlign (+ fused 2x2/s1 mean when ``avg``); differentiable w.r.t. ``feat`` only."""
    return _RoIAlignFn.apply(feat, rois, int(pooled_h), int(pooled_w), float(spatial_scale), bool(avg), bool(out_nchw))


class _RoIAlignSampledFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale, sampling, out_nchw):
        _need_cuda(feat, rois)
        rois = _rois_f32(rois)
        nhwc = _is_nhwc(feat)
        if not nhwc:
            feat = feat.contiguous()
        B, C, H, W = feat.shape
        R = rois.size(0)
        out = torch.empty((R, C, ph, pw), device=feat.device, dtype=torch.float32,
                          memory_format=torch.contiguous_format if out_nchw else _CL)
        check(lib.i2v_roi_align_sampled_fwd(ptr(feat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W, ptr(rois), R, ph, pw,
                                            scale, sampling, ptr(out), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, stream()),
              "roi_align_sampled_fwd")
        ctx.save_for_backward(rois)
        ctx.meta = (feat.shape, nhwc, ph, pw, scale, sampling, out_nchw)
        return out

    @staticmethod
    def backward(ctx, gout):
        (rois,) = ctx.saved_tensors
        shape, nhwc, ph, pw, scale, sampling, out_nchw = ctx.meta
        B, C, H, W = shape
        gout = gout.contiguous() if out_nchw else gout.contiguous(memory_format=_CL)
        gfeat = torch.empty(shape, device=gout.device, dtype=torch.float32,
                            memory_format=_CL if nhwc else torch.contiguous_format).zero_()
        check(lib.i2v_roi_align_sampled_bwd(ptr(gout), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, ptr(rois), rois.size(0), ph,
                                            pw, scale, sampling, ptr(gfeat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W,
                                            stream()), "roi_align_sampled_bwd")
        return gfeat, None, None, None, None, None, None


def roi_align_sampled(feat, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio=0, out_nchw=False):
    """``roi_layers.ROIAlign`` (maskrcnn-benchmark definition: mean of a sampling grid per bin); differentiable w.r.t.
    ``feat`` only, as roi_layers/roi_align.py:44 returns ``grad_input, None, ...``."""
    return _RoIAlignSampledFn.apply(feat, rois, int(pooled_h), int(pooled_w), float(spatial_scale), int(sampling_ratio),
                                    bool(out_nchw))


class _RoIPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale, out_nchw):
        _need_cuda(feat, rois)
        rois = _rois_f32(rois)
        nhwc = _is_nhwc(feat)
        if not nhwc:
            feat = feat.contiguous()
        B, C, H, W = feat.shape
        R = rois.size(0)
        fmt = torch.contiguous_format if out_nchw else _CL
        out = torch.empty((R, C, ph, pw), device=feat.device, dtype=torch.float32, memory_format=fmt)
        arg = torch.empty((R, C, ph, pw), device=feat.device, dtype=torch.int32, memory_format=fmt)
        check(lib.i2v_roi_pool_fwd(ptr(feat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W, ptr(rois), R, ph, pw,
                                   scale, ptr(out), ptr(arg), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, stream()),
              "roi_pool_fwd")
        ctx.save_for_backward(rois, arg)
        ctx.meta = (feat.shape, nhwc, ph, pw, out_nchw)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        rois, arg = ctx.saved_tensors
        shape, nhwc, ph, pw, out_nchw = ctx.meta
        B, C, H, W = shape
        gout = gout.contiguous() if out_nchw else gout.contiguous(memory_format=_CL)
        gfeat = torch.empty(shape, device=gout.device, dtype=torch.float32,
                            memory_format=_CL if nhwc else torch.contiguous_format).zero_()
        check(lib.i2v_roi_pool_bwd(ptr(gout), ptr(arg), LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, ptr(rois),
                                   rois.size(0), ph, pw, ptr(gfeat), LAYOUT_NHWC if nhwc else LAYOUT_NCHW, B, C, H, W,
                                   stream()), "roi_pool_bwd")
        return gfeat, None, None, None, None, None


def roi_pool(feat, rois, pooled_h, pooled_w, spatial_scale, out_nchw=True, return_argmax=False):
    out, arg = _RoIPoolFn.apply(feat, rois, int(pooled_h), int(pooled_w), float(spatial_scale), bool(out_nchw))
    return (out, arg) if return_argmax else out


class PackedMaps:
    """B feature maps (NHWC, packed: map b at b*H*W*C) at the start of a flat buffer of ``B * cap_cells * C`` floats whose
    extent (H, W) lives in DEVICE memory (``geom``: int32[2]).  What a captured step hands its ROI op when the size of the
    maps changes from batch to batch (roi_data_layer pads every batch to its own size): the recorded launch reads H and W
    at run time (``i2v_roi_pool_fwd_geom``)."""

    def __init__(self, buf, n_maps, channels, geom):
        _need_cuda(buf, geom)
        if geom.dtype != torch.int32 or geom.numel() < 2 or not geom.is_contiguous():
            raise ValueError("geom must be a contiguous int32 tensor [H, W]")
        if buf.dtype != torch.float32 or not buf.is_contiguous():
            raise ValueError("PackedMaps needs a contiguous fp32 buffer")
        self.buf, self.n_maps, self.channels, self.geom = buf, int(n_maps), int(channels), geom
        self.cap_cells = buf.numel() // (self.n_maps * self.channels)

    @property
    def device(self):
        return self.buf.device

    def view(self, h, w):
        """The maps as a (B,C,h,w) channels_last tensor (host-side knowledge of the extent: tests, eager code)."""
        n = self.n_maps * h * w * self.channels
        return self.buf[:n].view(self.n_maps, h, w, self.channels).permute(0, 3, 1, 2)


def roi_pool_packed(maps, rois, pooled_h, pooled_w, spatial_scale, out_nchw=True):
    """Caffe ROIPool over ``PackedMaps`` (forward only: the SGG_emb loop detaches the feature map,
    faster_rcnn_SGG_emb.py:148)."""
    rois = _rois_f32(rois)
    _need_cuda(rois)
    R, C = rois.size(0), maps.channels
    fmt = torch.contiguous_format if out_nchw else _CL
    out = torch.empty((R, C, pooled_h, pooled_w), device=rois.device, dtype=torch.float32, memory_format=fmt)
    arg = torch.empty((R, C, pooled_h, pooled_w), device=rois.device, dtype=torch.int32, memory_format=fmt)
    check(lib.i2v_roi_pool_fwd_geom(ptr(maps.buf), maps.n_maps, C, ptr(maps.geom), maps.cap_cells, ptr(rois), R,
                                    int(pooled_h), int(pooled_w), float(spatial_scale), ptr(out), ptr(arg),
                                    LAYOUT_NCHW if out_nchw else LAYOUT_NHWC, stream()), "roi_pool_fwd_geom")
    return out


# ----------------------------------------------------------------------------- NMS / RPN
def nms_sorted(dets, thresh, max_keep=0):
    """dets (n,5) or (n_img,n,5), rows in descending score order -> (keep int32 (n_img,n), num int32 (n_img,)),
    both on the device (no sync)."""
    _need_cuda(dets)
    d = dets.contiguous().float()
    if d.dim() == 2:
        d = d.unsqueeze(0)
    n_img, n = d.shape[0], d.shape[1]
    keep = torch.empty((n_img, max(n, 1)), device=d.device, dtype=torch.int32)
    num = torch.empty((n_img,), device=d.device, dtype=torch.int32)
    nb = lib.i2v_nms_workspace_bytes(n_img, n)
    ws = workspace(nb, d.device, "nms")
    check(lib.i2v_nms_sorted(ptr(d), n_img, n, float(thresh), int(max_keep), ptr(keep), ptr(num), ptr(ws), ws.numel(),
                             stream()), "nms_sorted")
    return keep, num


def sort_desc(keys):
    """(n_seg,n) fp32 -> int32 order (descending, ties by ascending index)."""
    _need_cuda(keys)
    k = keys.contiguous().float()
    if k.dim() == 1:
        k = k.unsqueeze(0)
    n_seg, n = k.shape
    order = torch.empty((n_seg, n), device=k.device, dtype=torch.int32)
    ws = workspace(lib.i2v_sort_desc_workspace_bytes(n_seg, n), k.device, "sort")
    check(lib.i2v_sort_desc(ptr(k), n_seg, n, ptr(order), ptr(ws), ws.numel(), stream()), "sort_desc")
    return order


def rpn_decode(cls_nhwc, bbox_nhwc, im_info, base_anchors, feat_stride, is_prob=False):
    """cls (B,2A,H,W) / bbox (B,4A,H,W) channels_last -> proposals (B,HWA,4), fg scores (B,HWA)."""
    _need_cuda(cls_nhwc, bbox_nhwc, im_info, base_anchors)
    cls_nhwc, bbox_nhwc = as_nhwc(cls_nhwc), as_nhwc(bbox_nhwc)
    B, C2, H, W = cls_nhwc.shape
    A = C2 // 2
    prop = torch.empty((B, H * W * A, 4), device=cls_nhwc.device, dtype=torch.float32)
    score = torch.empty((B, H * W * A), device=cls_nhwc.device, dtype=torch.float32)
    check(lib.i2v_rpn_decode(ptr(cls_nhwc), int(is_prob), ptr(bbox_nhwc), ptr(im_info.contiguous().float()),
                             ptr(base_anchors.contiguous().float()), B, H, W, A, int(feat_stride), ptr(prop), ptr(score),
                             stream()), "rpn_decode")
    return prop, score


def rpn_proposal(cls_nhwc, bbox_nhwc, im_info, base_anchors, feat_stride, pre_nms_top_n, post_nms_top_n, nms_thresh,
                 is_prob=False, want_index=False):
    """Whole proposal layer on the device -> rois (B,post,5) [, kept anchor idx (B,post), num (B,)]."""
    _need_cuda(cls_nhwc, bbox_nhwc, im_info, base_anchors)
    cls_nhwc, bbox_nhwc = as_nhwc(cls_nhwc), as_nhwc(bbox_nhwc)
    B, C2, H, W = cls_nhwc.shape
    A = C2 // 2
    dev = cls_nhwc.device
    rois = torch.empty((B, post_nms_top_n, 5), device=dev, dtype=torch.float32)
    kept = torch.empty((B, post_nms_top_n), device=dev, dtype=torch.int32) if want_index else None
    num = torch.empty((B,), device=dev, dtype=torch.int32) if want_index else None
    nb = lib.i2v_rpn_proposal_workspace_bytes(B, H, W, A, int(pre_nms_top_n))
    ws = workspace(nb, dev, "rpn")
    info = im_info.contiguous().float()
    base = base_anchors.contiguous().float()
    check(lib.i2v_rpn_proposal(ptr(cls_nhwc), int(is_prob), ptr(bbox_nhwc), ptr(info), ptr(base), B, H, W, A,
                               int(feat_stride), int(pre_nms_top_n), int(post_nms_top_n), float(nms_thresh), ptr(rois),
                               ptr(kept), ptr(num), ptr(ws), ws.numel(), stream()), "rpn_proposal")
    return (rois, kept, num) if want_index else rois


def bbox_overlaps(boxes, gt, want_matrix=False):
    """boxes (N,4) shared by all images or (B,N,4|5) [5: col0=batch idx]; gt (B,K,5).
    Returns (overlaps (B,N,K) or None, max (B,N), argmax (B,N) int32)."""
    _need_cuda(boxes, gt)
    gt = gt.contiguous().float()
    boxes = boxes.contiguous().float()
    B, K = gt.shape[0], gt.shape[1]
    batched = boxes.dim() == 3
    N = boxes.shape[-2]
    stride = boxes.shape[-1]
    off = 1 if stride == 5 else 0
    ov = torch.empty((B, N, K), device=gt.device, dtype=torch.float32) if want_matrix else None
    mx = torch.empty((B, N), device=gt.device, dtype=torch.float32)
    am = torch.empty((B, N), device=gt.device, dtype=torch.int32)
    check(lib.i2v_bbox_overlaps(ptr(boxes), stride, off, int(batched), ptr(gt), B, N, K, ptr(ov), ptr(mx), ptr(am),
                                stream()), "bbox_overlaps")
    return ov, mx, am


# ----------------------------------------------------------------------------- conv / linear
PROFILE = None      # bench.py sets this to a list: (start_event, end_event, flops, tag, desc, algorithmic bytes) per GEMM launch


class _Timed:
    """HIP events around one launch on the launch stream (torch's current stream) when profiling."""

    def __init__(self, flops, tag, desc="", nbytes=0):
        self.on = PROFILE is not None
        if self.on:
            self.rec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), flops, tag, desc,
                        float(nbytes))

    def __enter__(self):
        if self.on:
            self.rec[0].record()

    def __exit__(self, *a):
        if self.on:
            self.rec[1].record()
            PROFILE.append(self.rec)


class ZeroArena:
    """Pre-zeroed output arena for the split-K convolutions of one step: ONE clear per step instead of one
    hipMemsetAsync per launch (~70 per SGG_emb step).  ``reset()`` at the top of a step clears the prefix used
    last step and rewinds; ``take()`` hands out channels_last tensors.  When it runs out it reports the size
    it would have needed (``wanted``) and the conv falls back to clearing its own output."""

    def __init__(self, nbytes, device):
        self.buf = torch.zeros(max(int(nbytes), 1024) // 4, dtype=torch.float32, device=device)
        self.off = self.used = self.wanted = 0

    def reset(self):
        # while a graph is being captured the clear must cover everything the captured step will take, whatever the
        # eager steps before it used (an arena that was just re-sized has used == 0: a captured step without a clear
        # node would accumulate into its own outputs of the previous replay)
        n = self.buf.numel() if torch.cuda.is_current_stream_capturing() else self.used
        if n:
            self.buf[:n].zero_()
        self.off = self.wanted = 0

    def take(self, B, C, H, W):
        n = B * C * H * W
        n_al = (n + 63) // 64 * 64
        self.wanted += n_al
        if self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n].view(B, H, W, C).permute(0, 3, 1, 2)
        self.off += n_al
        self.used = max(self.used, self.off)
        return t

    def take_flat(self, n):
        """n zeros, contiguous (bias-gradient sums, small filter gradients that are accumulated atomically)."""
        n_al = (n + 63) // 64 * 64
        self.wanted += n_al
        if self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n_al
        self.used = max(self.used, self.off)
        return t


# Bumped whenever trained parameters change behind autograd's back (train.FusedSGD writes through raw device pointers and
# never touches ``Tensor._version``): part of the key of every cache derived from a trained parameter.
PARAM_EPOCH = 0


def param_key(w):
    return (w.data_ptr(), w._version, w.device, PARAM_EPOCH if getattr(w, "_i2v_trained", False) else 0)


SMALL_GW_BYTES = 16 << 20
ARENA = None        # set by a training step object (train.SGGEmbStep) around its forward/backward


class SplitWorkspace:
    """Caller-owned split-K scratch of the implicit-GEMM kernels (include/i2vsgg_hip.h, i2v_conv_fwd): arrival counters
    (zero between launches) + a slab of partial tiles.  Launches that share one must be ordered on the device, so every
    piece of work that may run CONCURRENTLY with another (two HIP graphs on two streams, two branches of one graph)
    gets its own: a step object installs it in ``ops.SPLIT_WS`` around the code it captures.  Without one, calls use a
    workspace per (device, launch stream)."""
    BYTES = (48 << 20) + 4096

    def __init__(self, device, nbytes=None):
        self.buf = torch.zeros(int(nbytes or self.BYTES), dtype=torch.uint8, device=device)


SPLIT_WS = None     # set like ARENA; None -> one workspace per launch stream
_TUNE_SPLIT_ATOMICS = 4                 # include/i2vsgg_hip.h I2V_TUNE_SPLIT_ATOMICS
ORDERED_SUMS = os.environ.get("I2V_ORDERED_SUMS", "1") != "0"     # 0: LaunchContext(ordered=True) is ignored (A/B of its cost)
_split_ws_by_stream = {}


class LaunchContext:
    """What one independently scheduled piece of captured work owns exclusively: the pre-zeroed arena of its
    atomically accumulated outputs, its split-K workspace and its tagged scratch buffers.  ``with ctx:`` installs them
    for the calls made inside (forward AND the autograd backward triggered inside the block)."""

    def __init__(self, device, arena=True, ordered=False):
        self.device = torch.device(device)
        self.arena = ZeroArena(1024, self.device) if arena else None      # sized after the first eager step (fit())
        self.split = SplitWorkspace(self.device)
        self.scratch = {}
        # ordered: every reduction the launches of this context split across workgroups -- split-K GEMMs of any split count,
        # small filter gradients, bias column sums -- is summed in a fixed order through ``split`` (I2V_TUNE_SPLIT_ATOMICS = 0
        # while the context is entered): bit-reproducible results.  The relation step's head asks for it (free there); the
        # instance_styleD step does not (+4 % of its step: DESIGN.md 5.10)
        self.ordered = bool(ordered)

    def fit(self):
        """After an eager step: re-size the arena to what the step asked for."""
        a = self.arena
        if a is not None and a.wanted * 4 > a.buf.numel() * 4:
            self.arena = ZeroArena(int(a.wanted * 4 * 1.05) + 4096, self.device)

    def __enter__(self):
        global ARENA, SPLIT_WS, SCRATCH
        self._saved = (ARENA, SPLIT_WS, SCRATCH)
        ARENA, SPLIT_WS, SCRATCH = self.arena, self.split, self.scratch
        self._tune = None
        if self.ordered and ORDERED_SUMS:
            self._tune = lib.i2v_get_tuning(_TUNE_SPLIT_ATOMICS)
            if self._tune == 2:             # an explicit I2V_SPLIT_ATOMICS=1 (always atomics) is the user's to keep
                lib.i2v_set_tuning(_TUNE_SPLIT_ATOMICS, 0)
        if self.arena is not None:
            self.arena.reset()          # one clear for every atomically accumulated output of this piece of work
        return self

    def __exit__(self, *exc):
        global ARENA, SPLIT_WS, SCRATCH
        ARENA, SPLIT_WS, SCRATCH = self._saved
        if self._tune == 2:
            lib.i2v_set_tuning(_TUNE_SPLIT_ATOMICS, 2)
        return False


# ---------------------------------------------------------------------------------------------------------------------
# Streams.  ``torch.cuda.Stream()`` does not create a stream: it deals the next of 32 pooled streams per device and priority,
# round robin -- and torch.cuda.graph's capture stream, ProcessGroupNCCL's streams and every caller's own streams come out of the
# same pool.  A process that has built a few step objects therefore holds "different" stream objects with the SAME handle, and
# a fork onto an alias of the forking stream (or of a sibling branch) is no fork at all.  The step objects take their streams
# from this registry instead: one HIP stream per (device, role), created ONCE per process by the library
# (``i2v_stream_create``: hipStreamCreateWithPriority, non-blocking), wrapped as a ``torch.cuda.ExternalStream`` and never
# destroyed.  Such a handle cannot come out of torch's pool, two roles never share one, and step objects built one after the
# other reuse the same few streams (a stream is an ordered queue: sharing a role between objects that run one after the other
# costs nothing).
_ROLE_STREAMS = {}
STREAM_REQUESTS = []    # every request in order (capped): with torch.cuda.Stream() each of them drew the next pooled handle


def role_stream(device, role, priority=0):
    """The process-wide stream of ``role`` (any hashable: "side", ("frame", 0), "copy", ...) on ``device``."""
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, role, int(priority))
    if len(STREAM_REQUESTS) < 4096:
        STREAM_REQUESTS.append(role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        import ctypes
        torch.cuda.init()
        out = ctypes.c_void_p()
        check(lib.i2v_stream_create(index, int(priority), ctypes.byref(out)), "i2v_stream_create")
        taken = {t.cuda_stream for t in _ROLE_STREAMS.values()}
        if not out.value or out.value in taken:
            raise _lib.I2VError("role_stream: the runtime handed out stream handle %r twice" % out.value)
        st = _ROLE_STREAMS[key] = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", index))
    return st


def stream_table():
    """{(device, role, priority): handle} of every stream the registry has created (tools/stream_handles.py, tests)."""
    return {k: v.cuda_stream for k, v in _ROLE_STREAMS.items()}


_BRANCH_DEPTH = 0
_FORKED = {}            # handle -> origin handle of every branch forked and not yet joined (ops.join)


class branch:
    """``with ops.branch(stream, origin):`` -- the body runs on ``stream`` as a fork of ``origin`` (stream.wait_stream(origin) first;
    the caller joins with ``ops.join(origin, stream, ...)``).  The step objects fork their graph branches through this so that the
    capture-time failures the schedule must avoid are error messages instead:
      * a fork made INSIDE a forked branch ends ``hipStreamEndCapture`` in a host segfault on ROCm 7.2 -- every
        branch forks from the capturing stream itself;
      * a branch stream whose HANDLE equals the origin's, or that of a sibling branch still open, is not a branch (the work is
        silently serialised, and events recorded "between" the two are edges of a stream onto itself)."""

    def __init__(self, stream, origin):
        self.stream, self.origin = stream, origin

    def __enter__(self):
        global _BRANCH_DEPTH
        if _BRANCH_DEPTH > 0 and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.branch: a fork inside a forked graph branch (hipStreamEndCapture crashes on it): fork every "
                               "branch from the capturing stream")
        h, ho = self.stream.cuda_stream, self.origin.cuda_stream
        if h == ho:
            raise RuntimeError("ops.branch: the branch stream IS the forking stream (handle %#x): take branch streams from "
                               "ops.role_stream, torch.cuda.Stream() deals pooled handles round robin" % h)
        if h in _FORKED:
            raise RuntimeError("ops.branch: stream %#x is already an open branch (a sibling's alias?); join it first" % h)
        self.stream.wait_stream(self.origin)
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        _FORKED[h] = ho
        _BRANCH_DEPTH += 1
        return self

    def __exit__(self, *exc):
        global _BRANCH_DEPTH
        _BRANCH_DEPTH -= 1
        if exc and exc[0] is not None:
            _FORKED.pop(self.stream.cuda_stream, None)       # a failed body: whoever handles the error owns the clean-up
        return self._ctx.__exit__(*exc)


def reset_branches():
    """After a failed capture: forget the branches it left open."""
    global _BRANCH_DEPTH
    _FORKED.clear()
    _BRANCH_DEPTH = 0


def join(origin, *streams):
    """``origin`` waits for every branch in ``streams`` (the join of ``ops.branch``)."""
    for st in streams:
        origin.wait_stream(st)
        _FORKED.pop(st.cuda_stream, None)


def _sws_args(device=None):
    """(pointer, bytes) of the split workspace in force: what the ordered cross-workgroup sums of round 5 (bias column sums,
    split filter gradients) need next to the split-K GEMMs."""
    t = _split_ws(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return ptr(t), t.numel()


def _split_ws(device):
    ws = SPLIT_WS
    if ws is None:
        key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
        ws = _split_ws_by_stream.get(key)
        if ws is None:
            ws = _split_ws_by_stream[key] = SplitWorkspace(device)
    return ws.buf


_side_ws = {}


def _side_split_ws(device):
    """The split workspace of the filter-gradient side branch (one per device: the branch is one ordered stream)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _side_ws.get(key)
    if ws is None:
        ws = _side_ws[key] = SplitWorkspace(device)
    return ws.buf


def _conv_fwd_raw(x, w, scale, shift, res, stride, pad, flags, out=None):
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = None
    sws = _split_ws(x.device)
    atomics = lib.i2v_conv_fwd_splits(B, H, W, Cin, Cout, KH, KW, stride, pad, sws.numel()) == 1
    if out is not None and not atomics:
        # the caller's buffer (a step's static feature-map buffer: no copy afterwards); only when the launch does not
        # accumulate into its output
        if tuple(out.shape) != (B, Cout, Ho, Wo) or not out.is_contiguous(memory_format=_CL) or out.dtype != torch.float32:
            raise ValueError("conv2d(out=...): needs a (%d,%d,%d,%d) channels_last fp32 tensor" % (B, Cout, Ho, Wo))
        y = out
    if y is None and ARENA is not None and atomics:
        y = ARENA.take(B, Cout, Ho, Wo)
        if y is not None:
            flags |= EPI_ZEROED
    if y is None:
        y = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
    # which kernel serves the call (profiling only): pointwise layers whose split-K, if any, finishes in the kernel run on
    # conv_gemm_f32, everything else on conv_igemm_f32 (csrc/conv.hip, launch_tile)
    kern = ""
    if PROFILE is not None:
        pointwise = (KH, KW, stride, pad) == (1, 1, 1, 0) and Cout % 4 == 0 and Cin % 4 == 0
        kern = " [gemm]" if pointwise and lib.i2v_get_tuning(10) and \
            lib.i2v_conv_fwd_splits(B, H, W, Cin, Cout, KH, KW, stride, pad, sws.numel()) == 0 else " [igemm]"
    with _Timed(2.0 * B * Ho * Wo * Cout * KH * KW * Cin, "fwd",
                "M%d N%d K%d (%dx%d s%d)%s" % (B * Ho * Wo, Cout, KH * KW * Cin, KH, KW, stride, kern),
                4 * (x.numel() + w.numel() + y.numel() + (res.numel() if res is not None else 0))):
        check(lib.i2v_conv_fwd(ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(res), ptr(y), B, H, W, Cin, Cout, KH, KW,
                               stride, pad, flags, ptr(sws), sws.numel(), stream()), "conv_fwd")
    if out is not None and y is not out:
        out.copy_(y)
        return out
    return y


LINEAR_DGRAD_AS_WGRAD = 0     # filter elements from which a linear layer's dgrad goes through the wgrad kernel


def _linear_dgrad_as_wgrad(in_shape, w_shape, stride, pad):
    B, Cin, H, W = in_shape
    Cout, _, KH, KW = w_shape
    return (H, W, KH, KW, stride, pad) == (1, 1, 1, 1, 1, 0) and B <= Cin and (Cout * Cin >= LINEAR_DGRAD_AS_WGRAD or Cout % 4 != 0)


def _conv_dgrad_raw(g, w, in_shape, stride, pad):
    B, Cin, H, W = in_shape
    Cout, _, KH, KW = w.shape
    dev = g.device
    if _linear_dgrad_as_wgrad(in_shape, w.shape, stride, pad):
        # linear layer: gx[m][k] = sum_n g[m][n] w[n][k] is a 'filter gradient' whose pixel axis is n, whose activations
        # are w as stored (n x k) and whose output gradient is g^T (n x m) -- only the small g is transposed, where the
        # implicit-GEMM form re-lays the whole filter out first (fc7: 134 -> 48 us in the step, the 64-row layers
        # 14-20 -> 10 us; Cout % 4 != 0: no zero-padded copies).  Only while g is the smaller of the two (rows <= in-features):
        # netD_style's 37500-row projections keep the implicit-GEMM form
        gt = g.reshape(B, Cout).t().contiguous().view(Cout, B, 1, 1)
        return _conv_wgrad_raw(w, gt, (B, Cin, 1, 1), 1, 0, tag="dgrad")
    if Cout % 4:
        # the transposed conv reduces over Cout: pad it to a float4 boundary with zero filters
        padc = 4 - Cout % 4
        g = torch.nn.functional.pad(g, (0, 0, 0, 0, 0, padc)).contiguous(memory_format=_CL)
        w = torch.cat([w, w.new_zeros((padc,) + tuple(w.shape[1:]))], 0).contiguous(memory_format=_CL)
        Cout += padc
    gx = torch.empty((B, Cin, H, W), device=dev, dtype=torch.float32, memory_format=_CL)
    ws = workspace(lib.i2v_conv_dgrad_workspace_bytes(Cin, Cout, KH, KW), dev, "dgrad")
    sws = _split_ws(dev)
    with _Timed(2.0 * B * g.shape[2] * g.shape[3] * Cout * KH * KW * Cin, "dgrad",
                "M%d N%d K%d" % (B * H * W, Cin, KH * KW * Cout), 4 * (g.numel() + w.numel() + gx.numel())):
        check(lib.i2v_conv_dgrad(ptr(g), ptr(w), ptr(gx), B, H, W, Cin, Cout, KH, KW, stride, pad, ptr(ws), ws.numel(),
                                 ptr(sws), sws.numel(), stream()), "conv_dgrad")
    return gx


# I2V_WINOGRAD_WGRAD=0: the filter gradient of a trained 3x3 layer stays on the direct kernel (its forward and data gradient
# are Winograd F(4x4,3x3) with WINOGRAD_TRAIN)
WINOGRAD_WGRAD = True


def _winograd_wgrad_ok(x, w_shape, stride, pad):
    Cout, Cin, KH, KW = w_shape
    return (WINOGRAD_WGRAD and WINOGRAD_TRAIN and (KH, KW, stride, pad) == (3, 3, 1, 1) and Cin % 4 == 0 and Cout % 4 == 0
            and Cin >= WINOGRAD_TRAIN_MIN_C and Cout >= WINOGRAD_TRAIN_MIN_C)


def _conv_wgrad_raw(x, g, w_shape, stride, pad, tag="wgrad", row_scale=None, winograd=False, v=None):
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w_shape
    # small filters: the pixel reduction is split over workgroups and accumulated with atomics, which needs a
    # zeroed gw -- take it from the step's pre-zeroed arena and accumulate (beta = 1) instead of one clear per call
    gw, beta = None, 0.0
    n = Cout * Cin * KH * KW
    # up to 224 pixels the launcher never splits the reduction (fewer than 8 stages of 32): one workgroup per tile
    # stores its result directly -- no zeroed output, no atomics
    if ARENA is not None and n * 4 <= SMALL_GW_BYTES and B * g.shape[2] * g.shape[3] > 224:
        flat = ARENA.take_flat(n)
        if flat is not None:
            gw, beta = flat.view(Cout, KH, KW, Cin).permute(0, 3, 1, 2), 1.0
    if gw is None:
        gw = torch.empty(w_shape, device=x.device, dtype=torch.float32, memory_format=_CL)
    if winograd and _winograd_wgrad_ok(x, w_shape, stride, pad):
        # stride-1 / pad-1 3x3 layer of a trained bottleneck: 36 plane GEMMs over the 4x4 tiles, a quarter of the MACs
        # a scratch buffer of its own: filter gradients may run on a side branch beside the Winograd data gradients
        ws = workspace(lib.i2v_conv3x3_winograd4_wgrad_workspace_bytes(B, H, W, Cin, Cout), x.device, "winograd_wgrad")
        T = B * ((H + 3) // 4) * ((W + 3) // 4)
        with _Timed(2.0 * B * H * W * Cout * 9 * Cin, tag, "N%d K%d M%d (3x3 winograd F4)" % (Cout, 9 * Cin, B * H * W),
                    4 * (x.numel() + g.numel() + gw.numel())):
            if v is not None:       # the forward's transformed input
                check(lib.i2v_conv3x3_winograd4_wgrad_v(ptr(v), ptr(g), ptr(row_scale), ptr(gw), B, H, W, Cin, Cout, beta, ptr(ws),
                                                        ws.numel(), stream()), "conv3x3_winograd4_wgrad_v")
            else:
                check(lib.i2v_conv3x3_winograd4_wgrad(ptr(x), ptr(g), ptr(row_scale), ptr(gw), B, H, W, Cin, Cout, beta, ptr(ws),
                                                      ws.numel(), stream()), "conv3x3_winograd4_wgrad")
        return gw
    with _Timed(2.0 * B * g.shape[2] * g.shape[3] * Cout * KH * KW * Cin, tag,
                "N%d K%d M%d" % (Cout, KH * KW * Cin, B * g.shape[2] * g.shape[3]) + (" (wgrad form)" if tag != "wgrad" else ""),
                4 * (x.numel() + g.numel() + gw.numel())):
        # ordered sum of a split reduction (bit-reproducible; no atomics, no clear) through the split workspace in force.  A call
        # on the filter-gradient side branch (InstanceStyleDStep.wgrad_branch) runs beside the main chain's GEMMs that own that
        # workspace -- launches sharing one must be ordered on the device -- so the side branch has a workspace of its own
        # (round 6; round 5 left its sums on atomics)
        side = WGRAD_STREAM is not None and torch.cuda.current_stream().cuda_stream == WGRAD_STREAM.cuda_stream
        sws = _side_split_ws(x.device) if side else _split_ws(x.device)
        if row_scale is not None:
            check(lib.i2v_conv_wgrad_scaled(ptr(x), ptr(g), ptr(row_scale), ptr(gw), B, H, W, Cin, Cout, KH, KW, stride, pad,
                                            beta, ptr(sws), sws.numel(), stream()), "conv_wgrad_scaled")
        else:
            check(lib.i2v_conv_wgrad(ptr(x), ptr(g), ptr(gw), B, H, W, Cin, Cout, KH, KW, stride, pad, beta,
                                     ptr(sws), sws.numel(), stream()), "conv_wgrad")
    return gw


# parameter storage pointer -> (momentum buffer, lr, momentum, weight_decay): filters whose SGD update is
# fused into their wgrad epilogue (train.FusedSGD.fuse_wgrad); their .grad is then never materialised
FUSED_SGD = {}


class FusedEntry(tuple):
    """(momentum buffer, lr, momentum, weight_decay) of one fused filter + the optimizer that registered it (``owner``)."""

    def __new__(cls, m, lr, momentum, wd, owner=None):
        self = super().__new__(cls, (m, lr, momentum, wd))
        self.owner = owner
        return self


def _conv_wgrad_sgd_raw(x, g, w, cfg, stride, pad):
    m, lr, mom, wd = cfg
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    with _Timed(2.0 * B * g.shape[2] * g.shape[3] * Cout * KH * KW * Cin, "wgrad",
                "N%d K%d M%d +sgd" % (Cout, KH * KW * Cin, B * g.shape[2] * g.shape[3]),
                4 * (x.numel() + g.numel() + 4 * w.numel())):        # filter and momentum each read and written once
        rc = lib.i2v_conv_wgrad_sgd(ptr(x), ptr(g), ptr(w), ptr(m), B, H, W, Cin, Cout, KH, KW, stride, pad, float(lr),
                                    float(mom), float(wd), stream())
    return rc


# I2V_WINOGRAD_TRAIN=0: direct kernels for trained 3x3 layers too (fp32 error 1e-6 instead of 1e-5 per layer)
WINOGRAD_TRAIN = True
WINOGRAD_TRAIN_MIN_C = 64


class _ConvFn(torch.autograd.Function):
    """y = relu?( conv(x,w)*scale + shift + res ).  scale/shift of a frozen BN get no gradient;
    a bias (shift without scale) does."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, res, stride, pad, relu, wino_ok=False):
        _need_cuda(x, w)
        x = as_nhwc(x)
        w = as_nhwc(w)
        flags = 0
        if scale is not None:
            flags |= EPI_SCALE
        elif shift is not None:
            flags |= EPI_BIAS
        if res is not None:
            res = as_nhwc(res)
            flags |= EPI_RESIDUAL
        if relu:
            flags |= EPI_RELU
        # a trained stride-1 / pad-1 3x3 layer whose caller allows it (the bottleneck 3x3s: instance_styleD trains
        # layer1-3 and layer4): forward and data gradient as Winograd F(4x4,3x3) with the filter transformed each step
        # (it changes each step); wgrad stays direct.  Not the RPN's 3x3: proposal ranking between near-tied scores
        # follows the conv's last bits, and the direct kernel's 1e-6 keeps 99 % of the reference's proposals, 1e-5 97 %
        wino = (wino_ok and WINOGRAD_TRAIN and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and res is None and stride == 1
                and pad == 1 and tuple(w.shape[2:]) == (3, 3)
                and w.shape[1] >= WINOGRAD_TRAIN_MIN_C and w.shape[0] >= WINOGRAD_TRAIN_MIN_C and w.shape[1] % 4 == 0)
        v = None
        if wino:
            keep = WINOGRAD_KEEP_V and ctx.needs_input_grad[1] and _winograd_wgrad_ok(x, w.shape, stride, pad)
            y, v = conv3x3_winograd(x, winograd_filter(w.detach(), 4), scale, shift, relu, keep_v=keep) if keep else \
                (conv3x3_winograd(x, winograd_filter(w.detach(), 4), scale, shift, relu), None)
        else:
            y = _conv_fwd_raw(x, w, scale, shift, res, stride, pad, flags)
        ctx.wino = wino
        # the data gradient of an eligible 3x3 layer takes the Winograd form even where the forward may not (the RPN conv)
        ctx.wino_dgrad = wino or (WINOGRAD_TRAIN and WINOGRAD_WGRAD and stride == 1 and pad == 1 and tuple(w.shape[2:]) == (3, 3)
                                  and w.shape[1] >= WINOGRAD_TRAIN_MIN_C and w.shape[0] >= WINOGRAD_TRAIN_MIN_C
                                  and w.shape[1] % 4 == 0 and w.shape[0] % 4 == 0)
        ctx.cfg = (stride, pad, relu, scale is not None, shift is not None, res is not None)
        ctx.save_for_backward(x, w, scale, y if relu else None, v)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, scale, y, v = ctx.saved_tensors
        stride, pad, relu, has_scale, has_shift, has_res = ctx.cfg
        gy = as_nhwc(gy)
        M, N = gy.shape[0] * gy.shape[2] * gy.shape[3], gy.shape[1]
        gres = g_t = None
        need_bias = ctx.needs_input_grad[3] and has_shift and not has_scale
        need_res = has_res and ctx.needs_input_grad[4]
        gbias = None
        if need_bias:
            gbias = ARENA.take_flat(N) if ARENA is not None else None
            if gbias is None:
                gbias = torch.zeros((N,), device=gy.device, dtype=torch.float32)

        # one pass over gy: g_pre = gy*(y>0) feeds the residual branch and the bias sum, g = g_pre*scale feeds
        # dgrad / wgrad
        if relu or has_scale or need_bias:
            want_pre = need_res and (relu or has_scale)
            want_g = has_scale or relu
            g = torch.empty_like(gy) if want_g else None
            gpre = torch.empty_like(gy) if (want_pre and has_scale) else None
            # a linear layer whose data gradient will run on the wgrad kernel wants g column-major as well: written by
            # this pass instead of a transpose kernel of its own
            if ctx.needs_input_grad[0] and not ctx.wino_dgrad and _linear_dgrad_as_wgrad(x.shape, w.shape, stride, pad):
                g_t = torch.empty((N, M, 1, 1), device=gy.device, dtype=torch.float32)
            check(lib.i2v_epilogue_bwd(ptr(gy), ptr(y), ptr(scale) if has_scale else None, ptr(g), ptr(gpre), ptr(gbias),
                                       M, N, int(relu), ptr(g_t), *_sws_args(), stream()), "epilogue_bwd")
            if g is None:
                g = gy
            if need_res:
                gres = gpre if gpre is not None else g       # without a BN scale g IS g_pre
        else:
            g = gy
            gres = gy if need_res else None
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.wino_dgrad:
                gx = conv3x3_winograd(g, winograd_filter_dgrad(w), tag="dgrad")
            elif g_t is not None:
                gx = _conv_wgrad_raw(w, g_t, (x.shape[0], x.shape[1], 1, 1), 1, 0, tag="dgrad")
            else:
                gx = _conv_dgrad_raw(g, w, x.shape, stride, pad)
        gw = None
        if ctx.needs_input_grad[1]:
            fused = FUSED_SGD.get(w.data_ptr())
            # the filter is read by dgrad above before it is updated here
            if fused is None or _conv_wgrad_sgd_raw(x, g, w, fused, stride, pad) != 0:
                # Winograd filter gradient for every eligible 3x3 (the RPN's too: its FORWARD stays direct for the proposal
                # ranking's sake, the filter gradient only feeds the next step's weights)
                gw = _conv_wgrad_raw(x, g, w.shape, stride, pad, winograd=True, v=v)
        return gx, gw, None, gbias, gres, None, None, None, None


# ---------------------------------------------------------------- a trained bottleneck as ONE autograd node
# I2V_BLOCK_FUSED=0: every conv of a trained bottleneck is its own autograd node again (one streaming pass over the
# activation gradient per conv for the BN scale / ReLU mask, the skip connection's gradient added by autograd)
BLOCK_FUSED = True


def _dgrad_fused(g, w, in_shape, pad, gy_scale=None, out_scale=None, res=None, mask=None, stride=1):
    """gx = mask > 0 ? (dgrad(g * gy_scale, w) * out_scale + res) : 0 of a stride-1 layer or a strided 1x1 layer (gx is then
    zero off the stride grid, and so must ``res`` be) (i2v_conv_dgrad_fused)."""
    B, Cin, H, W = in_shape
    Cout, _, KH, KW = w.shape
    gx = torch.empty((B, Cin, H, W), device=g.device, dtype=torch.float32, memory_format=_CL)
    ws = workspace(lib.i2v_conv_dgrad_workspace_bytes(Cin, Cout, KH, KW), g.device, "dgrad")
    sws = _split_ws(g.device)
    extra = (res.numel() if res is not None else 0) + (mask.numel() if mask is not None else 0)
    with _Timed(2.0 * B * g.shape[2] * g.shape[3] * Cout * KH * KW * Cin, "dgrad", "M%d N%d K%d +epi" % (B * H * W, Cin, KH * KW * Cout),
                4 * (g.numel() + w.numel() + gx.numel() + extra)):
        check(lib.i2v_conv_dgrad_fused(ptr(g), ptr(w), ptr(gy_scale), ptr(out_scale), ptr(res), ptr(mask), ptr(gx), B, H, W,
                                       Cin, Cout, KH, KW, stride, pad, ptr(ws), ws.numel(), ptr(sws), sws.numel(), stream()),
              "conv_dgrad_fused")
    return gx


def _wgrad_scaled(x, g, w_shape, pad, row_scale):
    """gw[n] = row_scale[n] * wgrad(x, g)[n] of a stride-1 layer; output placement as in _conv_wgrad_raw."""
    return _conv_wgrad_raw(x, g, w_shape, 1, pad, row_scale=row_scale)


def _winograd_dgrad_fused(g, U, out_scale, mask):
    """gx = mask > 0 ? (F(4x4,3x3) data gradient of a stride-1 / pad-1 3x3 layer * out_scale) : 0; U = winograd_filter_dgrad(w)."""
    B, Cout, H, W = g.shape
    Cin = U.shape[1]
    gx = torch.empty((B, Cin, H, W), device=g.device, dtype=torch.float32, memory_format=_CL)
    ws = workspace(lib.i2v_conv3x3_winograd4_workspace_bytes(B, H, W, Cout, Cin), g.device, "winograd")
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    with _Timed(2.0 * B * H * W * Cout * 9 * Cin, "dgrad",
                "M%d N%d K%d (3x3 winograd F4) gemmMB=%.2f +epi" % (B * H * W, Cin, 9 * Cout, 4e-6 * 36 * (T * Cout + Cout * Cin + T * Cin)),
                4 * (g.numel() + 9 * Cout * Cin + 2 * gx.numel())):
        check(lib.i2v_conv3x3_winograd4_dgrad(ptr(g), ptr(U), ptr(out_scale), ptr(mask), ptr(gx), B, H, W, Cout, Cin, ptr(ws),
                                              ws.numel(), stream()), "conv3x3_winograd4_dgrad")
    return gx


# Filter gradients of the block nodes on a side branch of the captured step (train.InstanceStyleDStep sets the stream): they
# depend on the data-gradient chain but nothing in the backward depends on them, so they can fill the chip beside it.  One
# edge per block (the side branch waits for the block's data gradients), one join before the gradient exchange.  The tensors
# a side launch reads are kept referenced until the join (WGRAD_PENDING): in a captured step a block freed on the main
# branch would otherwise be handed to a later main-branch allocation while the side branch still reads it.
WGRAD_STREAM = None
WGRAD_PENDING = []


def join_wgrad_branch():
    """Called by the step after backward(): the capturing stream waits for the filter-gradient branch."""
    if WGRAD_STREAM is not None:
        cur = torch.cuda.current_stream()
        if WGRAD_STREAM.cuda_stream == cur.cuda_stream:
            raise RuntimeError("join_wgrad_branch: the filter-gradient stream is the current stream")
        cur.wait_stream(WGRAD_STREAM)
    WGRAD_PENDING.clear()


class _BottleneckFn(torch.autograd.Function):
    """out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + skip(x)) of a STRIDE-1 bottleneck with frozen BNs
    (resnet_instance_styleD_bilinear.py:181-217) as one autograd node.  The forward is the four fused-epilogue convs of the
    layer-by-layer form.  The backward never makes a pass of its own over an activation gradient:

      * the BN scale and the ReLU mask between two convs ride in the epilogue of the data-gradient kernel that produces the
        gradient (out_scale, mask = the consumer conv's own input, which IS the ReLU output), or -- the scale in front of
        conv3 and the skip conv, where no ReLU sits -- in the transposed filter (dgrad) and in the per-filter-row factor of
        the filter gradient;
      * the skip connection's gradient is the residual operand of conv1's data gradient (autograd adds nothing);
      * ``in_relu``: x is the ReLU output of the previous block; the returned gx is then already multiplied by (x > 0), and
        ``out_premasked`` tells this block that ITS consumer did the same for it (only set when the next block is the only
        consumer of ``out``; model.faster_rcnn.layers.make_layer).
    Per block and step: 3 passes over inner activations + 4 over the block output + autograd's add (3) become 2 extra
    epilogue reads."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, s1, b1, s2, b2, s3, b3, sd, bd, in_relu, out_premasked, wino, stride=1):
        _need_cuda(x, w1, w2, w3)
        x = as_nhwc(x)
        w1, w2, w3 = as_nhwc(w1), as_nhwc(w2), as_nhwc(w3)
        a1 = _conv_fwd_raw(x, w1, s1, b1, None, stride, 0, EPI_SCALE | EPI_RELU)      # caffe style: the stride sits on conv1
        v2 = None
        if wino:
            a2 = conv3x3_winograd(a1, winograd_filter(w2.detach(), 4), s2, b2, True,
                                  keep_v=WINOGRAD_KEEP_V and _winograd_wgrad_ok(a1, w2.shape, 1, 1))
            if isinstance(a2, tuple):
                a2, v2 = a2
        else:
            a2 = _conv_fwd_raw(a1, w2, s2, b2, None, 1, 1, EPI_SCALE | EPI_RELU)
        if wd is not None:
            wd = as_nhwc(wd)
            res = _conv_fwd_raw(x, wd, sd, bd, None, stride, 0, EPI_SCALE)
        else:
            res = x
        out = _conv_fwd_raw(a2, w3, s3, b3, res, 1, 0, EPI_SCALE | EPI_RESIDUAL | EPI_RELU)
        ctx.flags = (bool(in_relu), bool(out_premasked), bool(wino), wd is not None)
        ctx.stride = int(stride)
        ctx.save_for_backward(x, a1, a2, out, w1, w2, w3, wd, s1, s2, s3, sd, v2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, a1, a2, out, w1, w2, w3, wd, s1, s2, s3, sd, v2 = ctx.saved_tensors
        in_relu, premasked, wino, has_ds = ctx.flags
        need = ctx.needs_input_grad
        g = as_nhwc(g)
        if premasked:
            gpre = g
        else:       # the consumers of ``out`` know nothing of its ReLU: one masking pass
            gpre = torch.empty_like(g)
            M, N = g.shape[0] * g.shape[2] * g.shape[3], g.shape[1]
            check(lib.i2v_epilogue_bwd(ptr(g), ptr(out), None, ptr(gpre), None, None, M, N, 1, None, *_sws_args(), stream()), "epilogue_bwd")
        st = ctx.stride
        # ---- data gradients (the chain the rest of the backward waits for)
        g2 = _dgrad_fused(gpre, w3, a2.shape, 0, gy_scale=s3, out_scale=s2, mask=a2)         # gradient at conv2's raw output
        if wino:
            g1 = _winograd_dgrad_fused(g2, winograd_filter_dgrad(w2), s1, a1)
        else:
            g1 = _dgrad_fused(g2, w2, a1.shape, 1, out_scale=s1, mask=a1)
        gx = None
        if need[0]:
            # a strided block: both 1x1 data gradients live on the stride grid (zero elsewhere), so the skip projection's can
            # be the residual operand of conv1's just as in the stride-1 case
            skip = _dgrad_fused(gpre, wd, x.shape, 0, gy_scale=sd, stride=st) if has_ds else gpre
            gx = _dgrad_fused(g1, w1, x.shape, 0, res=skip, mask=x if in_relu else None, stride=st)

        # ---- filter gradients (conv3: gy = gpre * s3 as a per-row factor), on the side branch when the step has one
        def wgrads():
            gw3 = _wgrad_scaled(a2, gpre, w3.shape, 0, s3) if need[3] else None
            gw2 = _conv_wgrad_raw(a1, g2, w2.shape, 1, 1, winograd=wino, v=v2) if need[2] else None
            gw1 = _conv_wgrad_raw(x, g1, w1.shape, st, 0) if need[1] else None
            gwd = _conv_wgrad_raw(x, gpre, wd.shape, st, 0, row_scale=sd) if (has_ds and need[4]) else None
            return gw1, gw2, gw3, gwd
        side = WGRAD_STREAM
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            WGRAD_PENDING.append((x, a1, a2, v2, gpre, g2, g1))
            with torch.cuda.stream(side):
                gw1, gw2, gw3, gwd = wgrads()
        else:
            gw1, gw2, gw3, gwd = wgrads()
        return (gx, gw1, gw2, gw3, gwd) + (None,) * 12


def bottleneck(x, w1, w2, w3, bn1, bn2, bn3, down=None, in_relu=False, out_premasked=False, stride=1):
    """A bottleneck with frozen BNs whose filters train, as one autograd node (``_BottleneckFn``).  bn* = (scale, shift) of the
    folded BN; ``down`` = (filter, scale, shift) of the projection on the skip branch or None; ``stride`` sits on conv1 and on
    the projection (caffe style), a strided block always has the projection."""
    if stride != 1 and down is None:
        raise ValueError("a strided bottleneck needs the projection on its skip branch")
    wino = (WINOGRAD_TRAIN and w2.shape[1] >= WINOGRAD_TRAIN_MIN_C and w2.shape[0] >= WINOGRAD_TRAIN_MIN_C and w2.shape[1] % 4 == 0)
    wd, sd, bd = down if down is not None else (None, None, None)
    return _BottleneckFn.apply(x, w1, w2, w3, wd, bn1[0], bn1[1], bn2[0], bn2[1], bn3[0], bn3[1], sd, bd, bool(in_relu),
                               bool(out_premasked), bool(wino), int(stride))


def conv2d(x, w, scale=None, shift=None, res=None, stride=1, pad=0, relu=False, winograd=False, out=None):
    """Implicit-GEMM conv with fused epilogue.  x (B,Cin,H,W), w (Cout,Cin,KH,KW); Cin % 4 == 0.
    ``out`` (no gradient being recorded only): the result is written into this channels_last tensor."""
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    if out is not None:
        if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
            raise ValueError("conv2d(out=...) is a forward-only path")
        flags = (EPI_SCALE if scale is not None else EPI_BIAS if shift is not None else 0) | (EPI_RESIDUAL if res is not None else 0) | \
            (EPI_RELU if relu else 0)
        _need_cuda(x, w)
        return _conv_fwd_raw(as_nhwc(x), as_nhwc(w), scale, shift, as_nhwc(res) if res is not None else None, int(stride), int(pad),
                             flags, out=out)
    if pad == 0 and KH == H and KW == W and (KH > 1 or KW > 1) and res is None:
        # the filter covers the whole input (vrd.conv_lo's 8x8 layer): one output pixel, i.e. a linear layer
        # over the NHWC-flattened map.  As a conv its dgrad is a full correlation with 63 of 64 taps masked.
        xf = as_nhwc(x).permute(0, 2, 3, 1).reshape(B, H * W * Cin, 1, 1)
        wf = w.contiguous(memory_format=_CL).permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin, 1, 1)
        return _ConvFn.apply(xf, wf, scale, shift, None, 1, 0, bool(relu))
    return _ConvFn.apply(x, w, scale, shift, res, int(stride), int(pad), bool(relu), bool(winograd))


def linear(x, w, b=None, relu=False):
    """nn.Linear (+ReLU) as a 1x1 conv over (M,1,1,K): x (M,K), w (N,K) -> (M,N)."""
    M, K = x.shape
    y = conv2d(x.contiguous().view(M, K, 1, 1), w.view(w.shape[0], K, 1, 1), None, b, None, 1, 0, relu)
    return y.view(M, w.shape[0])


def maxpool3x3s2(x):
    """Stem max pool: k3 s2 p0 ceil_mode (resnet_instance_styleD_bilinear.py:228).  No backward:
    the stem is frozen (RCNN_base[0..1] requires_grad=False, :392-393) and the pool input needs no grad
    unless layer1 inputs do -- handled by torch autograd through ``_MaxPoolFn``."""
    return _MaxPoolFn.apply(x)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = as_nhwc(x)
        B, C, H, W = x.shape
        Ho, Wo = (H - 3 + 1) // 2 + 1, (W - 3 + 1) // 2 + 1
        if (Ho - 1) * 2 >= H:
            Ho -= 1
        if (Wo - 1) * 2 >= W:
            Wo -= 1
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
        arg = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.int32, memory_format=_CL) \
            if x.requires_grad else None
        check(lib.i2v_maxpool3x3s2_fwd(ptr(x), ptr(y), ptr(arg), B, H, W, C, stream()), "maxpool3x3s2")
        ctx.shape = x.shape
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, gy):
        (arg,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        # scatter by saved argmax (plumbing; the stem is frozen on every path the reference trains)
        gx = torch.zeros((B, H * W, C), device=gy.device, dtype=torch.float32)
        gyf = gy.permute(0, 2, 3, 1).reshape(B, -1, C)
        idx = arg.permute(0, 2, 3, 1).reshape(B, -1, C).long()
        gx.scatter_add_(1, idx, gyf)
        return gx.view(B, H, W, C).permute(0, 3, 1, 2)


def sgd_momentum_(p, g, m, lr, momentum, weight_decay):
    """In-place fused SGD step on flat fp32 buffers."""
    _need_cuda(p, g, m)
    check(lib.i2v_sgd_momentum(ptr(p), ptr(g), ptr(m), p.numel(), float(lr), float(momentum), float(weight_decay),
                               stream()), "sgd_momentum")


def sgd_momentum_multi_(ps, gs, ms, lrs, wds, momentum):
    """The same step for many (small) tensors in one launch; all flat, contiguous fp32."""
    import ctypes
    n = len(ps)
    if n == 0:
        return
    _need_cuda(*ps, *gs, *ms)
    arr_p = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ps])
    arr_g = (ctypes.c_void_p * n)(*[t.data_ptr() for t in gs])
    arr_m = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ms])
    arr_n = (ctypes.c_int64 * n)(*[t.numel() for t in ps])
    arr_lr = (ctypes.c_float * n)(*[float(v) for v in lrs])
    arr_wd = (ctypes.c_float * n)(*[float(v) for v in wds])
    check(lib.i2v_sgd_momentum_multi(arr_p, arr_g, arr_m, arr_n, arr_lr, arr_wd, n, float(momentum), stream()),
          "sgd_momentum_multi")


def adam_multi_(ps, gs, ms, vs, lrs, wds, betas, eps, step_counter):
    """torch.optim.Adam's update for many tensors (flat, contiguous fp32) in as few launches as the table holds;
    ``step_counter``: the int32 device scalar ``adam_step_`` increments once per optimizer step."""
    import ctypes
    n = len(ps)
    if n == 0:
        return
    _need_cuda(*ps, *gs, *ms, *vs, step_counter)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    arr_n = (ctypes.c_int64 * n)(*[t.numel() for t in ps])
    arr_lr = (ctypes.c_double * n)(*[float(v) for v in lrs])
    arr_wd = (ctypes.c_float * n)(*[float(v) for v in wds])
    check(lib.i2v_adam_multi(arr(ps), arr(gs), arr(ms), arr(vs), arr_n, arr_lr, arr_wd, n, float(betas[0]), float(betas[1]),
                             float(eps), ptr(step_counter), stream()), "adam_multi")


def adam_step_(step_counter):
    check(lib.i2v_adam_step(ptr(step_counter), stream()), "adam_step")


class _DStylePoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, dim, rank):
        _need_cuda(x1, x2)
        x1, x2 = x1.contiguous(), x2.contiguous()
        n_img, rows, N = x1.shape
        z = torch.empty((n_img, dim), device=x1.device, dtype=torch.float32)
        check(lib.i2v_dstyle_pool_fwd(ptr(x1), ptr(x2), ptr(z), rows, n_img, dim, rank, stream()), "dstyle_pool_fwd")
        ctx.save_for_backward(x1, x2)
        ctx.dr = (dim, rank)
        return z

    @staticmethod
    def backward(ctx, gz):
        x1, x2 = ctx.saved_tensors
        dim, rank = ctx.dr
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        check(lib.i2v_dstyle_pool_bwd(ptr(gz.contiguous()), ptr(x1), ptr(x2), ptr(g1), ptr(g2), x1.shape[1], x1.shape[0],
                                      dim, rank, stream()), "dstyle_pool_bwd")
        return g1, g2, None, None


class _DStyleFusedFn(torch.autograd.Function):
    """netD_style's two projections + bilinear pooling as one kernel (i2v_dstyle_fused_fwd).  Forward-only calls keep
    nothing but z; when a gradient will be asked for, the projections are written as well and the backward is the
    existing chain on them: g1 = gz*x2, g2 = gz*x1 (one pass), then for each projection the bias column sums, the data
    gradient and the filter gradient of a linear layer."""

    @staticmethod
    def forward(ctx, rows, w1, b1, w2, b2, n_img, dim, rank):
        _need_cuda(rows, w1, b1, w2, b2)
        rows, w1, w2 = rows.contiguous(), w1.contiguous(), w2.contiguous()
        M, K = rows.shape
        P = M // n_img
        N = dim * rank
        dev = rows.device
        keep = any(ctx.needs_input_grad[:5])
        z = torch.empty((n_img, dim), device=dev, dtype=torch.float32)
        x1 = torch.empty((M, N), device=dev, dtype=torch.float32) if keep else None
        x2 = torch.empty((M, N), device=dev, dtype=torch.float32) if keep else None
        ws = workspace(lib.i2v_dstyle_fused_workspace_bytes(P, n_img, dim, rank), dev, "dstyle")
        with _Timed(2.0 * 2 * M * N * K, "fwd", "dstyle fused M%d N2x%d K%d" % (M, N, K),
                    4 * (rows.numel() + 2 * w1.numel() + (2 * M * N if keep else 0))):
            check(lib.i2v_dstyle_fused_fwd(ptr(rows), ptr(w1), ptr(b1.contiguous()), ptr(w2), ptr(b2.contiguous()), ptr(z),
                                           ptr(x1), ptr(x2), P, n_img, K, dim, rank, ptr(ws), ws.numel(), stream()),
                  "dstyle_fused_fwd")
        ctx.save_for_backward(rows, w1, w2, x1, x2)
        ctx.cfg = (n_img, dim, rank)
        return z

    @staticmethod
    def backward(ctx, gz):
        rows, w1, w2, x1, x2 = ctx.saved_tensors
        n_img, dim, rank = ctx.cfg
        M, K = rows.shape
        N = dim * rank
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        check(lib.i2v_dstyle_pool_bwd(ptr(gz.contiguous()), ptr(x1), ptr(x2), ptr(g1), ptr(g2), M // n_img, n_img, dim, rank,
                                      stream()), "dstyle_pool_bwd")
        rows4 = rows.view(M, K, 1, 1)
        grows = gw1 = gb1 = gw2 = gb2 = None
        outs = []
        for g, w, need_w, need_b in ((g1, w1, ctx.needs_input_grad[1], ctx.needs_input_grad[2]),
                                    (g2, w2, ctx.needs_input_grad[3], ctx.needs_input_grad[4])):
            g4 = g.view(M, N, 1, 1)
            gb = None
            if need_b:
                gb = ARENA.take_flat(N) if ARENA is not None else None
                if gb is None:
                    gb = torch.zeros((N,), device=g.device, dtype=torch.float32)
                check(lib.i2v_epilogue_bwd(ptr(g), None, None, None, None, ptr(gb), M, N, 0, None, *_sws_args(), stream()), "epilogue_bwd")
            gw = _conv_wgrad_raw(rows4, g4, (N, K, 1, 1), 1, 0).view(N, K) if need_w else None
            gx = _conv_dgrad_raw(g4, w.view(N, K, 1, 1), (M, K, 1, 1), 1, 0).view(M, K) if ctx.needs_input_grad[0] else None
            outs.append((gx, gw, gb))
        if ctx.needs_input_grad[0]:
            grows = outs[0][0].add_(outs[1][0])
        return grows, outs[0][1], outs[0][2], outs[1][1], outs[1][2], None, None, None


def dstyle_fused(rows, w1, b1, w2, b2, n_img, dim, rank):
    """z (n_img, dim) of netD_style from the (n_img * positions, 512) rows of the style tap: both 512 -> dim*rank
    projections and the product / rank / spatial reduction in one kernel."""
    return _DStyleFusedFn.apply(rows, w1, b1, w2, b2, int(n_img), int(dim), int(rank))


def dstyle_pool(x1, x2, dim, rank):
    """z[b,d] = sum_pos sum_r x1[b,pos,d*rank+r]*x2[b,pos,d*rank+r] (x1,x2: (B,rows,dim*rank))."""
    return _DStylePoolFn.apply(x1, x2, int(dim), int(rank))


class _DPixelFn(torch.autograd.Function):
    """netD_pixel as one kernel per direction (i2v_dpixel_fwd / _bwd) + three 1x1 filter gradients."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, lamb, pix_per_roi, want_feat):
        _need_cuda(x, w1, w2, w3)
        M = x.shape[0]
        dev = x.device
        h1 = torch.empty((M, 512), device=dev, dtype=torch.float32)
        h2 = torch.empty((M, 128), device=dev, dtype=torch.float32)
        d = torch.empty((M,), device=dev, dtype=torch.float32)
        feat = torch.empty((M // pix_per_roi, 128), device=dev, dtype=torch.float32) if want_feat else None
        check(lib.i2v_dpixel_fwd(ptr(x), ptr(w1), ptr(w2), ptr(w3), ptr(h1), ptr(h2), ptr(d), ptr(feat), M, pix_per_roi,
                                 stream()), "dpixel_fwd")
        ctx.save_for_backward(x, w1, w2, w3, h1, h2, d)
        ctx.cfg = (float(lamb), int(pix_per_roi), want_feat)
        return d, feat

    @staticmethod
    def backward(ctx, gd, gfeat):
        x, w1, w2, w3, h1, h2, d = ctx.saved_tensors
        lamb, pix, want_feat = ctx.cfg
        M, dev = x.shape[0], x.device
        gd = gd.contiguous() if gd is not None else None
        gfeat = gfeat.contiguous() if (want_feat and gfeat is not None) else None
        g3 = torch.empty((M,), device=dev, dtype=torch.float32)
        gh2 = torch.empty((M, 128), device=dev, dtype=torch.float32)
        gh1 = torch.empty((M, 512), device=dev, dtype=torch.float32)
        gx = torch.empty((M, 1024), device=dev, dtype=torch.float32)
        ws = workspace(lib.i2v_dpixel_bwd_workspace_bytes(), dev, "dpixel")
        check(lib.i2v_dpixel_bwd(ptr(gd), ptr(gfeat), ptr(d), ptr(h1), ptr(h2), ptr(w1), ptr(w2), ptr(w3), ptr(g3), ptr(gh2),
                                 ptr(gh1), ptr(gx), M, pix, lamb, ptr(ws), ws.numel(), stream()), "dpixel_bwd")
        as4 = lambda t: t.view(M, -1, 1, 1)
        gw1 = _conv_wgrad_raw(as4(x), as4(gh1), (512, 1024, 1, 1), 1, 0).view(512, 1024) if ctx.needs_input_grad[1] else None
        gw2 = _conv_wgrad_raw(as4(h1), as4(gh2), (128, 512, 1, 1), 1, 0).view(128, 512) if ctx.needs_input_grad[2] else None
        gw3 = _conv_wgrad_raw(as4(h2), as4(g3), (1, 128, 1, 1), 1, 0).view(128) if ctx.needs_input_grad[3] else None
        return (gx if ctx.needs_input_grad[0] else None), gw1, gw2, gw3, None, None, None


def dpixel(x_rows, w1, w2, w3, lamb=1.0, pix_per_roi=49, want_feat=False):
    """Fused instance discriminator.  x_rows (M,1024) ROI pixels in NHWC order; w1 (512,1024), w2 (128,512), w3 (128).
    Returns (d (M,), feat (M/pix_per_roi,128) or None).  The gradient w.r.t. x is reversed and scaled by ``lamb``."""
    d, feat = _DPixelFn.apply(x_rows.contiguous(), w1.contiguous(), w2.contiguous(), w3.contiguous(), float(lamb),
                              int(pix_per_roi), bool(want_feat))
    return d, feat


def detection_postprocess(rois, cls_prob, bbox_pred, im_h, im_w, im_scale, class_agnostic=False, stds=None, means=None,
                          score_thresh=0.0, nms_thresh=0.3, max_per_image=100, im_info=None, out=None):
    """Per-class detection post-processing of one image on the device (test_net_instance_styleD_bilinear.py:151-221).

    rois (R,5) [batch_idx,x1,y1,x2,y2]; cls_prob (R,C); bbox_pred (R,4) or (R,4C).  ``stds`` / ``means``: the
    TRAIN.BBOX_NORMALIZE_STDS / _MEANS 4-tuples when the deltas are normalised, else None.  ``im_info``: the frame's
    [height, width, scale] as a device tensor of 3 floats, read by the kernel instead of the three host numbers (a
    captured step: the same launch for every frame); ``out``: (dets, counts) buffers to write into.
    Returns (dets (C,R,5), counts (C,) int32), both on the device: rows ``dets[j, :counts[j]]`` are the reference's
    ``all_boxes[j][i]``."""
    import ctypes
    _need_cuda(rois, cls_prob, bbox_pred)
    rois = rois.reshape(-1, 5).float().contiguous()
    cls_prob = cls_prob.reshape(rois.shape[0], -1).float().contiguous()
    R, C = cls_prob.shape
    bbox_pred = bbox_pred.reshape(R, -1).float().contiguous()
    if bbox_pred.shape[1] != (4 if class_agnostic else 4 * C):
        raise ValueError("bbox_pred has %d columns, expected %d" % (bbox_pred.shape[1], 4 if class_agnostic else 4 * C))
    dev = rois.device
    if out is not None:
        dets, counts = out
        if tuple(dets.shape) != (C, R, 5) or tuple(counts.shape) != (C,) or dets.dtype != torch.float32 or \
                counts.dtype != torch.int32 or not dets.is_contiguous():
            raise ValueError("detection_postprocess(out=...): needs (%d,%d,5) float32 and (%d,) int32 buffers" % (C, R, C))
    else:
        dets = torch.empty((C, R, 5), device=dev, dtype=torch.float32)
        counts = torch.empty((C,), device=dev, dtype=torch.int32)
    ws = workspace(lib.i2v_det_postprocess_workspace_bytes(R, C), dev, "det")
    f4 = lambda v: (ctypes.c_float * 4)(*[float(t) for t in v]) if v is not None else None
    if im_info is not None:
        _need_cuda(im_info)
        if im_info.dtype != torch.float32 or im_info.numel() != 3 or not im_info.is_contiguous():
            raise ValueError("detection_postprocess(im_info=...): 3 contiguous floats on the device")
        check(lib.i2v_det_postprocess_info(ptr(rois), ptr(cls_prob), ptr(bbox_pred), int(bool(class_agnostic)), f4(stds), f4(means),
                                           ptr(im_info), R, C, float(score_thresh), float(nms_thresh), int(max_per_image),
                                           ptr(dets), ptr(counts), ptr(ws), ws.numel(), stream()), "det_postprocess_info")
        return dets, counts
    check(lib.i2v_det_postprocess(ptr(rois), ptr(cls_prob), ptr(bbox_pred), int(bool(class_agnostic)), f4(stds), f4(means),
                                  float(im_h), float(im_w), float(im_scale), R, C, float(score_thresh), float(nms_thresh),
                                  int(max_per_image), ptr(dets), ptr(counts), ptr(ws), ws.numel(), stream()),
          "det_postprocess")
    return dets, counts


def image_prep(img_u8, pixel_means, target_size, flipped=False, rgb=True, blob=None):
    """Data-layer front-end of one image on the device (minibatch.py:60-90 + blob.py:35-52): img_u8 (H,W,3) uint8 on
    the device, file channel order RGB (``rgb=True``) or BGR.  Writes the mean-subtracted, resized BGR image into
    ``blob`` ((1,4,Hb,Wb) channels_last fp32, zero-initialised; allocated to fit when None -- channel 3 is the stem's
    zero pad) and returns (blob, (Ho, Wo, im_scale))."""
    import ctypes
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError("img_u8 must be a (H,W,3) uint8 CUDA tensor")
    img_u8 = img_u8.contiguous()
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    ho, wo, sc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_float()
    check(lib.i2v_image_prep_size(H, W, int(target_size), ctypes.byref(ho), ctypes.byref(wo), ctypes.byref(sc)), "image_prep_size")
    if blob is None:
        blob = torch.zeros((1, 4, ho.value, wo.value), device=img_u8.device, dtype=torch.float32).contiguous(memory_format=_CL)
    if blob.dim() != 4 or blob.shape[1] != 4 or not blob.is_contiguous(memory_format=_CL):
        raise ValueError("blob must be a (1,4,H,W) channels_last tensor")
    means = (ctypes.c_float * 3)(*[float(v) for v in np.asarray(pixel_means).reshape(-1)[:3]])
    check(lib.i2v_image_prep(ptr(img_u8), H, W, int(bool(rgb)), int(bool(flipped)), means, int(target_size), ptr(blob),
                             int(blob.shape[2]), int(blob.shape[3]), stream()), "image_prep")
    return blob, (ho.value, wo.value, float(sc.value))


def relation_topk(rel_score, conf, ixs, ixo, k=100):
    """Top-k (pair, predicate) cells of ``rel_score * conf[ixs] * conf[ixo]`` (lib/utils.py:584-628), on the device.
    rel_score (n_pairs, n_rel) fp32, conf (n_boxes,) fp32, ixs / ixo (n_pairs,) int64.  Returns (pair, pred, conf):
    int32, int32, fp32 tensors of length min(k, n_pairs * n_rel), descending."""
    _need_cuda(rel_score, conf, ixs, ixo)
    rel_score, conf = rel_score.float().contiguous(), conf.float().contiguous()
    ixs, ixo = ixs.long().contiguous(), ixo.long().contiguous()
    n_pairs, n_rel = rel_score.shape
    k = min(int(k), n_pairs * n_rel)
    dev = rel_score.device
    pair = torch.empty((k,), device=dev, dtype=torch.int32)
    pred = torch.empty((k,), device=dev, dtype=torch.int32)
    out = torch.empty((k,), device=dev, dtype=torch.float32)
    ws = workspace(lib.i2v_relation_topk_workspace_bytes(n_pairs, n_rel), dev, "reltopk")
    check(lib.i2v_relation_topk(ptr(rel_score), ptr(conf), ptr(ixs), ptr(ixo), n_pairs, n_rel, k, ptr(pair), ptr(pred), ptr(out),
                                ptr(ws), ws.numel(), stream()), "relation_topk")
    return pair, pred, out


class _L2NormRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        _need_cuda(x)
        x = x.contiguous()
        rows, cols = x.shape
        y = torch.empty_like(x)
        inv = torch.empty((rows,), device=x.device, dtype=torch.float32)
        check(lib.i2v_l2norm_rows_fwd(ptr(x), ptr(y), ptr(inv), rows, cols, float(eps), stream()), "l2norm_rows_fwd")
        ctx.save_for_backward(y, inv)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(g)
        check(lib.i2v_l2norm_rows_bwd(ptr(g), ptr(y), ptr(inv), ptr(gx), y.shape[0], y.shape[1], ctx.eps, stream()),
              "l2norm_rows_bwd")
        return gx, None


def l2norm_rows(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1) for a 2-D fp32 tensor, forward and backward one kernel each."""
    return _L2NormRowsFn.apply(x, eps)


class _BceRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, t, w):
        _need_cuda(z, t, w)
        z, t, w = z.contiguous(), t.contiguous(), w.contiguous()
        loss = torch.empty((), device=z.device, dtype=torch.float32)
        check(lib.i2v_bce_rows_fwd(ptr(z), ptr(t), ptr(w), ptr(loss), z.shape[0], z.shape[1], stream()), "bce_rows_fwd")
        ctx.save_for_backward(z, t, w)
        return loss

    @staticmethod
    def backward(ctx, gl):
        z, t, w = ctx.saved_tensors
        gz = torch.empty_like(z)
        check(lib.i2v_bce_rows_bwd(ptr(z), ptr(t), ptr(w), ptr(gl.contiguous()), ptr(gz), z.shape[0], z.shape[1], stream()),
              "bce_rows_bwd")
        return gz, None, None


def bce_rows(z, t, w):
    """sum_r w[r] * mean_c BCEWithLogits(z[r,c], t[r,c]) as one device scalar (forward and backward one kernel each; the
    forward is one workgroup with a fixed summation order: no clear, no atomics)."""
    return _BceRowsFn.apply(z, t, w)


class _PairGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, ixs, ixo):
        _need_cuda(obj, ixs, ixo)
        obj, ixs, ixo = obj.contiguous(), ixs.contiguous(), ixo.contiguous()
        nb, E = obj.shape
        out = torch.empty((ixs.numel(), 2 * E), device=obj.device, dtype=torch.float32)
        check(lib.i2v_pair_gather_fwd(ptr(obj), ptr(ixs), ptr(ixo), ptr(out), ixs.numel(), nb, E, stream()), "pair_gather_fwd")
        ctx.save_for_backward(ixs, ixo)
        ctx.shape = (nb, E)
        return out

    @staticmethod
    def backward(ctx, g):
        ixs, ixo = ctx.saved_tensors
        nb, E = ctx.shape
        gobj = torch.empty((nb, E), device=g.device, dtype=torch.float32)
        check(lib.i2v_pair_gather_bwd(ptr(g.contiguous()), ptr(ixs), ptr(ixo), ptr(gobj), ixs.numel(), nb, E, stream()),
              "pair_gather_bwd")
        return gobj, None, None


def pair_gather(obj, ixs, ixo):
    """[obj[ixs] | obj[ixo]] per relation pair: (n_pairs, 2*emb) from obj (n_box, emb), ixs / ixo int64 -- one kernel each way
    (the aten form is index_select, permute + copy forward; copy, zeros, index_add_ backward), deterministic backward."""
    if ixs.dtype != torch.long or ixo.dtype != torch.long:
        raise ValueError("pair_gather needs int64 indices")
    return _PairGatherFn.apply(obj, ixs, ixo)


class _HalfMseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, target):
        _need_cuda(d)
        d = d.contiguous()
        out = torch.empty((), device=d.device, dtype=torch.float32)
        check(lib.i2v_half_mse_fwd(ptr(d), d.numel(), float(target), ptr(out), stream()), "half_mse_fwd")
        ctx.save_for_backward(d)
        ctx.target = float(target)
        return out

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gd = torch.empty_like(d)
        check(lib.i2v_half_mse_bwd(ptr(d), d.numel(), ctx.target, ptr(g.contiguous()), ptr(gd), stream()), "half_mse_bwd")
        return gd, None


def half_mse(d, target=0.0):
    """0.5 * mean((d - target)^2) as one kernel each way: the discriminator terms of
    trainval_net_instance_styleD_bilinear.py:276-296 (0.5*mean(d**2), 0.5*mean((1-d)**2))."""
    return _HalfMseFn.apply(d, target)


class _SmoothL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, inw, outw, sigma, rows):
        _need_cuda(pred, tgt, inw, outw)
        pred, tgt, inw, outw = pred.contiguous(), tgt.contiguous(), inw.contiguous(), outw.contiguous()
        n = pred.numel()
        if tgt.numel() != n or inw.numel() != outw.numel() or n % inw.numel():
            raise ValueError("smooth_l1: shapes %s %s %s %s" % (tuple(pred.shape), tuple(tgt.shape), tuple(inw.shape), tuple(outw.shape)))
        per = n // inw.numel()
        out = torch.empty((), device=pred.device, dtype=torch.float32)
        check(lib.i2v_smooth_l1_fwd(ptr(pred), ptr(tgt), ptr(inw), ptr(outw), n, per, int(rows), float(sigma), ptr(out), stream()),
              "smooth_l1_fwd")
        ctx.save_for_backward(pred, tgt, inw, outw)
        ctx.cfg = (per, int(rows), float(sigma))
        return out

    @staticmethod
    def backward(ctx, g):
        pred, tgt, inw, outw = ctx.saved_tensors
        per, rows, sigma = ctx.cfg
        gp = torch.empty_like(pred)
        check(lib.i2v_smooth_l1_bwd(ptr(pred), ptr(tgt), ptr(inw), ptr(outw), pred.numel(), per, rows, sigma, ptr(g.contiguous()),
                                    ptr(gp), stream()), "smooth_l1_bwd")
        return gp, None, None, None, None, None


def smooth_l1(pred, tgt, inw, outw, sigma=1.0):
    """net_utils._smooth_l1_loss (net_utils.py:122-136) for the two call sites of the detector: the weights either match pred
    element for element or carry one value per trailing group (``(B,N,1)`` against ``(B,N,4)``); summed over everything but
    axis 0, averaged over axis 0.  Gradient w.r.t. pred only.  One kernel each way."""
    return _SmoothL1Fn.apply(pred, tgt, inw, outw, float(sigma), int(pred.shape[0]))


def bbox_transform(ex, gt, means=None, stds=None):
    """bbox_transform_batch (bbox_transform.py:36-75) in one kernel: ex (N,4) or (B,N,4), gt (B,N,>=4) with the box in its first
    four columns -> (B,N,4); ``means`` / ``stds`` (4 floats each): the normalised targets of
    proposal_target_layer_cascade.py:104-106.  No gradient."""
    import ctypes
    _need_cuda(ex, gt)
    ex = ex.contiguous().float()
    gt = gt.float()
    if gt.stride(-1) != 1 or gt.dim() != 3 or gt.stride(1) != gt.shape[2] or gt.stride(0) != gt.shape[1] * gt.shape[2]:
        gt = gt.contiguous()
    B, N = gt.shape[0], gt.shape[1]
    if ex.shape[-2] != N or ex.shape[-1] != 4:
        raise ValueError("bbox_transform: ex %s against gt %s" % (tuple(ex.shape), tuple(gt.shape)))
    out = torch.empty((B, N, 4), device=gt.device, dtype=torch.float32)
    f4 = lambda v: (ctypes.c_float * 4)(*[float(t) for t in v]) if v is not None else None
    check(lib.i2v_bbox_transform(ptr(ex), int(ex.dim() == 3), ptr(gt), int(gt.shape[2]), ptr(out), B, N, f4(means), f4(stds),
                                 stream()), "bbox_transform")
    return out


class _SignedSqrtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        _need_cuda(z)
        z = z.contiguous()
        y = torch.empty_like(z)
        check(lib.i2v_signed_sqrt_fwd(ptr(z), ptr(y), z.numel(), stream()), "signed_sqrt_fwd")
        ctx.save_for_backward(z)
        return y

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        gz = torch.empty_like(z)
        check(lib.i2v_signed_sqrt_bwd(ptr(z), ptr(g.contiguous()), ptr(gz), z.numel(), stream()), "signed_sqrt_bwd")
        return gz


def signed_sqrt(z):
    """sqrt(relu(z)) - sqrt(relu(-z)) (netD_style, resnet_instance_styleD_bilinear.py:137) as one kernel each way."""
    return _SignedSqrtFn.apply(z)


def winograd_filter(w, m=2):
    """(Cout,Cin,3,3) filter -> Winograd-domain filter, done once for a frozen filter: (16,Cout,Cin) for F(2x2,3x3)
    (``m=2``), (36,Cout,Cin) for F(4x4,3x3) (``m=4``)."""
    _need_cuda(w)
    Cout, Cin, KH, KW = w.shape
    if (KH, KW) != (3, 3) or m not in (2, 4):
        raise ValueError("winograd_filter needs a 3x3 filter and m in (2, 4)")
    w = w.contiguous(memory_format=_CL)
    U = torch.empty(((m + 2) ** 2, Cout, Cin), device=w.device, dtype=torch.float32)
    fn = lib.i2v_winograd_filter if m == 2 else lib.i2v_winograd4_filter
    check(fn(ptr(w), ptr(U), Cout, Cin, stream()), "winograd_filter")
    return U


def winograd_filter_dgrad(w):
    """(Cout,Cin,3,3) filter -> (36,Cin,Cout) Winograd F(4x4,3x3) filter of the layer's DATA gradient (taps flipped,
    channels swapped): ``conv3x3_winograd(gy, winograd_filter_dgrad(w))`` is dgrad of the stride-1 / pad-1 layer."""
    _need_cuda(w)
    Cout, Cin, KH, KW = w.shape
    if (KH, KW) != (3, 3):
        raise ValueError("winograd_filter_dgrad needs a 3x3 filter")
    w = w.contiguous(memory_format=_CL)
    U = torch.empty((36, Cin, Cout), device=w.device, dtype=torch.float32)
    check(lib.i2v_winograd4_filter_dgrad(ptr(w), ptr(U), Cout, Cin, stream()), "winograd_filter_dgrad")
    return U


# I2V_WINOGRAD_KEEP_V=0: a trained 3x3 layer transforms its input again for the filter gradient instead of keeping the forward's
# transformed input (36/16 of the activation's size per layer, ~1.9 GB over the 33 layers of an 8-frame instance_styleD step)
WINOGRAD_KEEP_V = True


def conv3x3_winograd(x, U, scale=None, shift=None, relu=False, tag="fwd", keep_v=False):
    """stride-1 / pad-1 3x3 convolution with a pre-transformed frozen filter U (ops.winograd_filter); forward only.
    ``keep_v`` (F(4x4,3x3) only): -> (y, V), V = the transformed input in a tensor of its own (for the filter gradient)."""
    _need_cuda(x, U)
    x = as_nhwc(x)
    B, Cin, H, W = x.shape
    Cout = U.shape[1]
    y = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32, memory_format=_CL)
    four = U.shape[0] == 36          # F(4x4,3x3)
    wsb = (lib.i2v_conv3x3_winograd4_workspace_bytes if four else lib.i2v_conv3x3_winograd_workspace_bytes)(B, H, W, Cin, Cout)
    ws = workspace(wsb, x.device, "winograd")
    fn = lib.i2v_conv3x3_winograd4_fwd if four else lib.i2v_conv3x3_winograd_fwd
    # the batched GEMM of the call is one conv_gemm_f32 launch over the planes: its operand bytes, for the traffic roofline
    planes, tl = (36, 4) if four else (16, 2)
    T = B * ((H + tl - 1) // tl) * ((W + tl - 1) // tl)
    gemm_mb = 4e-6 * planes * (T * Cin + Cout * Cin + T * Cout)
    with _Timed(2.0 * B * H * W * Cout * 9 * Cin, tag,
                "M%d N%d K%d (3x3 winograd F%d) gemmMB=%.2f" % (B * H * W, Cout, 9 * Cin, 4 if four else 2, gemm_mb),
                4 * (x.numel() + 9 * Cout * Cin + y.numel())):
        if keep_v and four:
            v = torch.empty((lib.i2v_conv3x3_winograd4_v_bytes(B, H, W, Cin) // 4,), device=x.device, dtype=torch.float32)
            check(lib.i2v_conv3x3_winograd4_fwd_keep(ptr(x), ptr(U), ptr(scale), ptr(shift), ptr(y), ptr(v), B, H, W, Cin, Cout,
                                                     int(bool(relu)), ptr(ws), ws.numel(), stream()), "conv3x3_winograd4_fwd_keep")
            return y, v
        check(fn(ptr(x), ptr(U), ptr(scale), ptr(shift), ptr(y), B, H, W, Cin, Cout, int(bool(relu)), ptr(ws), ws.numel(),
                 stream()), "conv3x3_winograd_fwd")
    return (y, None) if keep_v else y
