#!/usr/bin/env python3
"""Uninitialised-read hunt: fill the caching allocator's free blocks with NaN, then run the trained-bottleneck stack on
the fused and on the layer-by-layer path and compare with float64 torch.  A NaN (or a run-to-run difference) means some
kernel read memory it never wrote."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402
from i2vsgg_amd.model.faster_rcnn.layers import make_layer  # noqa: E402

DEV = "cuda:0"


def poison(gb=6):
    junk = [torch.full((256 << 20,), float("nan"), device=DEV) for _ in range(gb)]
    small = [torch.full((n,), float("nan"), device=DEV) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(8)]
    torch.cuda.synchronize()
    del junk, small


def reference(layer, x):
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


NB = int(os.environ.get("NB", "3"))
MODES = [m == "f" for m in os.environ.get("MODES", "fp")]
POISON = os.environ.get("POISON", "1") == "1"
for cin, planes, hw in [(512, 128, (14, 20))]:
    torch.manual_seed(3)
    layer, _ = make_layer(cin, planes, NB, 1)
    layer = layer.to(DEV)
    for m in layer.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 2.0); m.running_mean.normal_(0, 0.2)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.invalidate()
    x0 = torch.relu(torch.randn(2, cin, *hw, device=DEV)).contiguous(memory_format=torch.channels_last)
    gout = torch.randn(2, planes * 4, *hw, device=DEV).contiguous(memory_format=torch.channels_last)
    params = [p for p in layer.parameters() if p.requires_grad]
    names = ["out", "gx"] + [n for n, p in layer.named_parameters() if p.requires_grad]

    def run(fn, x0=x0):
        x = x0.clone().requires_grad_(True)
        for p in params:
            p.grad = None
        y = fn(x)
        (y * gout.to(y.dtype)).sum().backward()
        return [y.detach().float().clone(), x.grad.float().clone()] + [p.grad.clone() for p in params]

    ref = run(lambda x: reference(layer, x), x0.double())
    for rep in range(int(os.environ.get("REPS", "3"))):
        for fusedflag in MODES:
            if POISON:
                poison()
            ops.BLOCK_FUSED = fusedflag
            got = run(layer)
            ops.BLOCK_FUSED = True
            bad = []
            for n, a, c in zip(names, got, ref):
                scale = float(c.abs().max()) + 1e-12
                err = float((a - c).abs().max()) if bool(torch.isfinite(a).all()) else float("nan")
                if not err <= 1e-3 * scale:
                    bad.append("%s err %.3g/%.3g nan %d" % (n, err, scale, int((~torch.isfinite(a)).sum())))
            print("C%d rep %d %-5s: %s" % (cin, rep, "fused" if fusedflag else "plain", "ok" if not bad else "; ".join(bad[:6])), flush=True)
