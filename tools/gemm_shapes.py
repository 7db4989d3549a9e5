#!/usr/bin/env python3
"""The product's pointwise GEMM (i2v_conv_fwd, 1x1) on the shapes the round-6 review names, alone and as three co-running
chains (three streams: what a step's three graph branches look like to a CU), for the cost model's tile and for every forced
tile.  One HIP-event pair around `reps` back-to-back launches behind a blocker GEMM (alone); for the co-run the three streams
start behind one event and the slowest stream's end is the time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402
from i2vsgg_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
CHAINS = int(os.environ.get("CHAINS", "3"))        # co-running copies of the launch (the relation step: 3 branches; the detector step: 2)
TILES = ["128x128", "128x64", "96x64", "80x64", "64x64", "32x64"]
blocker = torch.randn(8192, 8192, device=DEV)


def alone(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(blocker, blocker)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def corun(fns, reps):
    streams = [torch.cuda.Stream() for _ in fns]
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in fns]
        torch.mm(blocker, blocker)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for _ in range(reps):
            for s, fn in zip(streams, fns):
                with torch.cuda.stream(s):
                    fn()
        for s, e in zip(streams, ends):
            e.record(s)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(e) for e in ends) * 1e3 / (reps * len(fns)))
    return best


cases = [("4096^3", 4096, 4096, 4096, False, 10), ("layer3 conv1, 4 frames", 9576, 1024, 256, False, 30),
         ("layer3 conv3, 2 frames (+res)", 4788, 256, 1024, True, 30), ("layer3 conv1, 2 frames", 4788, 1024, 256, False, 30),
         ("layer3 conv3, 1 frame (+res)", 2394, 256, 1024, True, 30), ("layer3 conv1, 1 frame", 2394, 1024, 256, False, 30),
         ("layer2 conv3, 1 frame (+res)", 9375, 128, 512, True, 30), ("layer2 conv1, 1 frame", 9375, 512, 128, False, 30),
         # the stage-split step runs every backbone kernel over BOTH frames of a minibatch (round 6)
         ("layer2 conv3, 2 frames (+res)", 18750, 128, 512, True, 30), ("layer2 conv1, 2 frames", 18750, 512, 128, False, 30),
         ("layer1 conv3, 2 frames (+res)", 75000, 64, 256, True, 30), ("layer1 conv1, 2 frames", 75000, 256, 64, False, 30),
         # configs[2]: a domain's 4 frames per launch, two branches (CHAINS=2)
         ("layer3 conv3, 4 frames (+res)", 9576, 256, 1024, True, 30),
         ("layer2 conv3, 4 frames (+res)", 37500, 128, 512, True, 20), ("layer2 conv1, 4 frames", 37500, 512, 128, False, 20),
         ("layer1 conv3, 4 frames (+res)", 150000, 64, 256, True, 20), ("layer1 conv1, 4 frames", 150000, 256, 64, False, 20),
         ("layer4 conv3, 128 rois 7x7 (+res)", 6272, 512, 2048, True, 30), ("layer4 conv1, 128 rois 7x7", 6272, 2048, 512, False, 30),
         # the relation head's fc6 forward: 128 rows against the 822 MB filter (streams from HBM), in an ordered context as in the step
         ("fc6 forward (ordered ctx)", 128, 50176, 4096, False, 10)]
only = sys.argv[1:]
for name, M, K, N, res, reps in cases:
    if only and not any(o in name for o in only):
        continue
    fl = 2.0 * M * N * K
    ops_ = []
    for c in range(CHAINS):
        # uniform [-1, 1) operands, the convention of /opt/skills/guides (what the lab uses too): the clock the chip holds
        # depends on the data, so rates measured on other distributions do not compare
        x = torch.rand(M, K, 1, 1, device=DEV) * 2 - 1
        w = torch.rand(N, K, 1, 1, device=DEV) * 2 - 1
        sc, sh = torch.rand(N, device=DEV) + 0.5, torch.rand(N, device=DEV)
        r = torch.randn(M, N, 1, 1, device=DEV) if res else None
        ctx = ops.LaunchContext(DEV, ordered="ordered" in name)
        ops_.append((x, w, sc, sh, r, ctx))

    def make(c):
        x, w, sc, sh, r, ctx = ops_[c]

        def fn():
            with ctx, torch.no_grad():
                ops.conv2d(x, w, sc, sh, r, 1, 0, relu=True)
        return fn
    fns = [make(c) for c in range(CHAINS)]
    print("== %s  M %d K %d N %d (%.2f GFLOP)" % (name, M, K, N, fl * 1e-9), flush=True)
    for t in [-1] + list(range(len(TILES))):
        lib.i2v_conv_set_tile(t if t >= 0 else -1)
        a = alone(fns[0], reps)
        c3 = corun(fns, reps)
        print("   tile %-8s alone %8.2f us %6.1f TF | %d chains %8.2f us/launch %6.1f TF" % (
            TILES[t] if t >= 0 else "model", a, fl / a / 1e6, CHAINS, c3, fl / c3 / 1e6), flush=True)
    lib.i2v_conv_set_tile(-1)
