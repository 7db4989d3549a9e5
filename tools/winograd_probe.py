#!/usr/bin/env python3
"""Winograd F(2x2,3x3) vs the direct implicit-GEMM kernel on the backbone's 3x3 shapes: error and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops
B = 2
for name, c, h, w in (("l1 c2 3x3 64", 64, 150, 250), ("l2 c2 3x3 128", 128, 75, 125), ("l3 c2 3x3 256", 256, 38, 63),
                      ("l4 c2 3x3 512 (64 rois)", 512, 7, 7)):
    b = 64 if h == 7 else B
    x = torch.randn(b, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, device="cuda") * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(c, device="cuda") + 0.5, torch.rand(c, device="cuda") - 0.5
    U = ops.winograd_filter(wt)
    ref = ops.conv2d(x, wt, sc, sh, None, 1, 1, relu=True)
    got = ops.conv3x3_winograd(x, U, sc, sh, relu=True)
    ref64 = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    e_d = ((ref.double() - ref64).abs().max() / ref64.abs().max()).item()
    e_w = ((got.double() - ref64).abs().max() / ref64.abs().max()).item()
    def t(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    td = t(lambda: ops.conv2d(x, wt, sc, sh, None, 1, 1, relu=True))
    tw = t(lambda: ops.conv3x3_winograd(x, U, sc, sh, relu=True))
    U4 = ops.winograd_filter(wt, 4)
    got4 = ops.conv3x3_winograd(x, U4, sc, sh, relu=True)
    e_4 = ((got4.double() - ref64).abs().max() / ref64.abs().max()).item()
    t4 = t(lambda: ops.conv3x3_winograd(x, U4, sc, sh, relu=True))
    print("%-26s direct %6.1f us (err %.1e)   F(2x2) %6.1f us (err %.1e)   F(4x4) %6.1f us (err %.1e)" % (name, td, e_d, tw, e_w, t4, e_4))

# ---- pieces of the layer3 case: input transform / batched GEMM / output transform
import ctypes
from i2vsgg_amd._lib import lib
from i2vsgg_amd.ops import ptr, stream
c, h, w = 256, 38, 63
x = torch.randn(2, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
T = 2 * 19 * 32
V = torch.randn(16, T, c, device="cuda"); U = torch.randn(16, c, c, device="cuda") * 0.05; M = torch.empty(16, T, c, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tg = t(lambda: lib.i2v_gemm_nt_batched(ptr(V), ptr(U), ptr(M), T, c, c, 16, T * c, c * c, T * c, None, 0, stream()))
print("batched GEMM 16 x (%d x %d x %d): %.1f us = %.1f TF" % (T, c, c, tg, 16 * 2.0 * T * c * c / tg / 1e6))
for tile in range(6):
    lib.i2v_conv_set_tile(tile)
    tg = t(lambda: lib.i2v_gemm_nt_batched(ptr(V), ptr(U), ptr(M), T, c, c, 16, T * c, c * c, T * c, None, 0, stream()))
    print("   tile %d: %.1f us" % (tile, tg))
lib.i2v_conv_set_tile(-1)
