#!/usr/bin/env python3
"""What a plain streaming kernel reaches on the tensor sizes of the HBM-bound pointwise layers (copy, add(a, c) -> b, in-place ReLU):
the bound a fused 64 -> 256 / 128 -> 512 expansion with residual can approach (its bytes are add's plus the small input)."""
import torch

dev = "cuda:0"


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for M, N in ((75000, 256), (300000, 256), (75000, 512), (18750, 512)):
    a, c = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev)
    b = torch.empty_like(a)
    t1, t2, t3 = timed(lambda: b.copy_(a)), timed(lambda: torch.add(a, c, out=b)), timed(lambda: torch.relu_(b))
    by = M * N * 4
    print("M%d N%d (%.0f MB): copy %.1f us = %.2f TB/s | add(a, c) -> b %.1f us = %.2f TB/s | relu_ %.1f us = %.2f TB/s" % (
        M, N, by / 1e6, t1 * 1e6, 2 * by / t1 / 1e12, t2 * 1e6, 3 * by / t2 / 1e12, t3 * 1e6, 2 * by / t3 / 1e12))
