#!/usr/bin/env python3
"""Linear-layer data gradient two ways: (a) i2v_conv_dgrad (filter re-layout pass + NT implicit GEMM), (b) the wgrad
kernel with the roles swapped -- gx[m][k] = sum_n gy[m][n] w[n][k] is a 'filter gradient' whose pixel axis is n, whose
activations are w (n x k, as stored) and whose output gradient is gy^T (n x m): only the small gy is transposed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((128, 4096, 4096), (64, 300, 4096), (64, 256, 4096), (64, 256, 600), (64, 256, 64), (64, 256, 768), (64, 300, 256),
                (62, 1024, 300), (62, 300, 1024), (64, 64, 8192), (64, 62, 300)):
    g = torch.randn(M, N, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    a = lambda: ops._conv_dgrad_raw(g.view(M, N, 1, 1), w.view(N, K, 1, 1), (M, K, 1, 1), 1, 0)
    def b():
        gt = g.t().contiguous()
        return ops._conv_wgrad_raw(w.view(N, K, 1, 1), gt.view(N, M, 1, 1), (M, K, 1, 1), 1, 0)
    ra, rb = a().view(M, K), b().view(M, K)
    ref = g.double() @ w.double()
    ea = ((ra.double() - ref).abs().max() / ref.abs().max()).item()
    eb = ((rb.double() - ref).abs().max() / ref.abs().max()).item()
    print("M%-4d N%-5d K%-5d  dgrad %6.1f us (err %.1e)   wgrad-form %6.1f us (err %.1e)" % (M, N, K, t(a), ea, t(b), eb))
