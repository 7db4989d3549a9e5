#!/usr/bin/env python3
"""Which op of a small bottleneck (2 x 14 x 20 pixels, 512 / 128 channels) is not reproducible?  Each op 200 times on the
same inputs, NaN-poisoned allocator in between, against its float64 value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
B, H, W = 2, int(os.environ.get("H", 14)), int(os.environ.get("W", 20))
C4, C1 = 512, 128
x512 = cl(torch.randn(B, C4, H, W, device=DEV)); x128 = cl(torch.randn(B, C1, H, W, device=DEV))
g512 = cl(torch.randn(B, C4, H, W, device=DEV)); g128 = cl(torch.randn(B, C1, H, W, device=DEV))
w1 = cl(torch.randn(C1, C4, 1, 1, device=DEV) * 0.05); w2 = cl(torch.randn(C1, C1, 3, 3, device=DEV) * 0.05)
w3 = cl(torch.randn(C4, C1, 1, 1, device=DEV) * 0.05)
s128, b128 = torch.rand(C1, device=DEV) + 0.5, torch.randn(C1, device=DEV)
s512, b512 = torch.rand(C4, device=DEV) + 0.5, torch.randn(C4, device=DEV)
d = lambda t: t.double()


def poison():
    junk = [torch.full((n,), float("nan"), device=DEV) for n in (1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(6)]
    del junk


CASES = {
    "conv1 fwd 512->128": (lambda: ops._conv_fwd_raw(x512, w1, s128, b128, None, 1, 0, ops.EPI_SCALE | ops.EPI_RELU),
                           lambda: F.relu(F.conv2d(d(x512), d(w1)) * d(s128).view(1, -1, 1, 1) + d(b128).view(1, -1, 1, 1))),
    "conv3 fwd 128->512+res": (lambda: ops._conv_fwd_raw(x128, w3, s512, b512, x512, 1, 0, ops.EPI_SCALE | ops.EPI_RELU | ops.EPI_RESIDUAL),
                               lambda: F.relu(F.conv2d(d(x128), d(w3)) * d(s512).view(1, -1, 1, 1) + d(b512).view(1, -1, 1, 1) + d(x512))),
    "conv2 wino fwd": (lambda: ops.conv3x3_winograd(x128, ops.winograd_filter(w2, 4), s128, b128, True),
                       lambda: F.relu(F.conv2d(d(x128), d(w2), None, 1, 1) * d(s128).view(1, -1, 1, 1) + d(b128).view(1, -1, 1, 1))),
    "conv3 dgrad 512->128": (lambda: ops._conv_dgrad_raw(g512, w3, x128.shape, 1, 0), lambda: F.conv_transpose2d(d(g512), d(w3))),
    "conv1 dgrad 128->512": (lambda: ops._conv_dgrad_raw(g128, w1, x512.shape, 1, 0), lambda: F.conv_transpose2d(d(g128), d(w1))),
    "conv2 wino dgrad": (lambda: ops.conv3x3_winograd(g128, ops.winograd_filter_dgrad(w2), tag="dgrad"),
                         lambda: F.conv_transpose2d(d(g128), d(w2), None, 1, 1)),
    "conv3 wgrad": (lambda: ops._conv_wgrad_raw(x128, g512, w3.shape, 1, 0), lambda: torch.nn.grad.conv2d_weight(d(x128), w3.shape, d(g512))),
    "conv1 wgrad": (lambda: ops._conv_wgrad_raw(x512, g128, w1.shape, 1, 0), lambda: torch.nn.grad.conv2d_weight(d(x512), w1.shape, d(g128))),
    "conv2 wgrad 3x3": (lambda: ops._conv_wgrad_raw(x128, g128, w2.shape, 1, 1), lambda: torch.nn.grad.conv2d_weight(d(x128), w2.shape, d(g128), 1, 1)),
    "epilogue_bwd": (lambda: ops.conv2d(x128.clone().requires_grad_(True), w3, s512, b512, None, 1, 0, relu=True).sum() * 0 + 0, None),
}
N = int(os.environ.get("N", 200))
for name, (fn, ref) in CASES.items():
    if ref is None:
        continue
    want = ref().float()
    scale = float(want.abs().max())
    worst, nbad = 0.0, 0
    for i in range(N):
        if i % 10 == 0:
            poison()
        got = fn()
        err = float((got - want).abs().max())
        worst = max(worst, err) if err == err else float("nan")
        nbad += not (err <= 1e-3 * scale)
    print("%-26s %d runs: worst err %.3g (scale %.3g)  bad runs %d" % (name, N, worst, scale, nbad), flush=True)
