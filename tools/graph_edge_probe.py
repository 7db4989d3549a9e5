#!/usr/bin/env python3
"""Do event edges between the branches of a captured graph cost concurrency?  Two chains of 40 convs (one frame of layer3
conv3) as branches of one graph: independent, vs chain B's i-th launch waiting for an event recorded behind chain A's i-th
launch (the shape of 'filter gradients on a side branch, fed by the data-gradient chain')."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
N = 40
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
xs = [cl(torch.randn(1, 256, 38, 63, device=DEV)) for _ in range(2)]
wt = cl(torch.randn(1024, 256, 1, 1, device=DEV) * 0.05)
sc, sh = torch.rand(1024, device=DEV) + 0.5, torch.rand(1024, device=DEV)
ctxs = [ops.LaunchContext(DEV) for _ in range(2)]
for x, ctx in zip(xs, ctxs):
    for _ in range(2):
        with ctx:
            ops.conv2d(x, wt, sc, sh, None, 1, 0, relu=True)
    ctx.fit()
torch.cuda.synchronize()
for mode in ("independent", "edges every launch", "edges every 8th launch", "one chain of 80"):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        if mode == "one chain of 80":
            for i in range(2 * N):
                with ctxs[0]:
                    ops.conv2d(xs[0], wt, sc, sh, None, 1, 0, relu=True)
        else:
            side.wait_stream(cap)
            every = 1 if mode == "edges every launch" else 8 if mode == "edges every 8th launch" else 0
            for i in range(N):
                with ctxs[0]:
                    ops.conv2d(xs[0], wt, sc, sh, None, 1, 0, relu=True)
                if every and i % every == 0:
                    ev = torch.cuda.Event()
                    ev.record(cap)
                    side.wait_event(ev)
                with torch.cuda.stream(side):
                    with ctxs[1]:
                        ops.conv2d(xs[1], wt, sc, sh, None, 1, 0, relu=True)
            cap.wait_stream(side)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-24s %7.1f us per replay (80 launches)" % (mode, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
    del g
