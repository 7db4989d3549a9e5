#!/usr/bin/env python3
"""HBM-side micro-benchmarks at the north-star sizes: ROIAlignAvg fwd/bwd, ROIPool, RPN proposal layer, NMS.
Prints achieved GB/s against the ALGORITHMIC bytes of SURVEY.md 8d (feature map read once + output written once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from i2vsgg_amd import ops, synthetic as syn
from i2vsgg_amd.model.rpn.generate_anchors import generate_anchors

dev = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for B, R in ((1, 32), (1, 128), (4, 32)):
    feat = torch.randn(B, 1024, 38, 63, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    rois = np.zeros((B * R, 5), np.float32)
    for b in range(B):
        rois[b * R:(b + 1) * R, 0] = b
        rois[b * R:(b + 1) * R, 1:] = syn.boxes(b, R)
    rt = torch.from_numpy(rois).to(dev)
    fbytes = B * 1024 * 38 * 63 * 4
    obytes = B * R * 1024 * 49 * 4
    t = timeit(lambda: ops.roi_align(feat.detach(), rt, 7, 7, 1 / 16.0, avg=True))
    print("ROIAlignAvg fwd  B=%d R=%3d/frame: %7.1f us  algorithmic %6.2f MB -> %7.1f GB/s" % (B, R, t * 1e6, (fbytes + obytes) / 1e6, (fbytes + obytes) / t / 1e9))
    out = ops.roi_align(feat, rt, 7, 7, 1 / 16.0, avg=True)
    g = torch.randn_like(out)
    def bwd():
        feat.grad = None
        out.backward(g, retain_graph=True)
    t = timeit(bwd)
    print("ROIAlignAvg bwd  B=%d R=%3d/frame: %7.1f us  algorithmic %6.2f MB -> %7.1f GB/s (incl. zero-fill)" % (B, R, t * 1e6, (fbytes + obytes) / 1e6, (fbytes + obytes) / t / 1e9))
    t = timeit(lambda: ops.roi_pool(feat.detach(), rt, 7, 7, 1 / 16.0, out_nchw=True))
    print("ROIPool     fwd  B=%d R=%3d/frame: %7.1f us  algorithmic %6.2f MB -> %7.1f GB/s" % (B, R, t * 1e6, (fbytes + 2 * obytes) / 1e6, (fbytes + 2 * obytes) / t / 1e9))

for n in (6000, 12000):
    for clustered in (False, True):
        dets = torch.from_numpy(syn.tie_free_dets(n, n, clustered=clustered)).to(dev)
        for mk in (0, 300 if n == 6000 else 2000):
            t = timeit(lambda: ops.nms_sorted(dets, 0.7, mk))
            ref_bytes = 20 * n + 2 * 8 * n * ((n + 63) // 64)
            k = int(ops.nms_sorted(dets, 0.7, mk)[1])
            print("NMS n=%5d %s max_keep=%4d: %7.1f us  kept %5d  reference-algorithm bytes %5.1f MB -> %6.1f GB/s" % (
                n, "clustered" if clustered else "uniform  ", mk, t * 1e6, k, ref_bytes / 1e6, ref_bytes / t / 1e9))

base = torch.from_numpy(generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))).float().to(dev)
for B in (1, 2, 4):
    cls = torch.randn(B, 18, 38, 63, device=dev).contiguous(memory_format=torch.channels_last)
    box = (torch.randn(B, 36, 38, 63, device=dev) * 0.2).contiguous(memory_format=torch.channels_last)
    info = torch.tensor([[600, 1000, 1.0]] * B, device=dev)
    for mode, pre, post in (("train", 12000, 2000), ("test", 6000, 300), ("target", 12000, 32)):
        t = timeit(lambda: ops.rpn_proposal(cls, box, info, base, 16, pre, post, 0.7))
        print("proposal layer B=%d %-6s (decode+sort+NMS+pad, 21546 anchors/frame): %7.1f us = %6.1f us/frame" % (B, mode, t * 1e6, t * 1e6 / B))

# ---- per-class detection post-processing of one image (300 rois), device pass vs the numpy
#      restatement of the reference's loop (1 + (C-1) host NMS calls) on this box's host
import time
from oracle import rpn as orpn
for C in (16, 36):
    R = 300
    rng = np.random.default_rng(C)
    xy = rng.uniform(0, 1, (R, 2)) * [880, 480]
    wh = rng.uniform(16, 300, (R, 2))
    rois = np.concatenate([np.zeros((R, 1)), xy, np.minimum(xy + wh, [999, 599])], 1).astype(np.float32)
    logits = rng.standard_normal((R, C)).astype(np.float32) * 2
    prob = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    pred = (rng.standard_normal((R, 4 * C)) * 0.5).astype(np.float32)
    args = (600.0, 1000.0, 1.6, False, (0.1, 0.1, 0.2, 0.2), (0.0, 0.0, 0.0, 0.0), 0.0, 0.3, 100)
    d = [torch.from_numpy(a).to(dev) for a in (rois, prob, pred)]
    t = timeit(lambda: ops.detection_postprocess(*d, *args))
    t0 = time.perf_counter()
    for _ in range(3):
        orpn.detection_postprocess(rois, prob, pred, *args)
    tc = (time.perf_counter() - t0) / 3
    print("detection post-processing R=300 C=%2d (decode + %2d x {threshold, sort, NMS 0.3} + top-100): %6.1f us on the device; "
          "numpy restatement of the reference loop %7.1f us on the host" % (C, C - 1, t * 1e6, tc * 1e6))
