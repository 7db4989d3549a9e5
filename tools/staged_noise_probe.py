#!/usr/bin/env python3
"""How far do the fc7 weights of tests/test_gpu_models.py::test_sgg_step_staged_batches_meet_their_features's two schedules
(eager one-pass / captured per-frame branches) differ from run to run, with the default-stream redirect on and off?
Prints max-abs relative differences: eager vs eager, graph vs eager (redirect on), graph vs eager (redirect off)."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = "cuda:0"
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
seeds = [3, 11, 12, 13, 14]


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


def run(graph):
    net = train.build_sgg_net(layers=50, seed=5, device=DEV)
    net.vrd.dropout = False
    step = train.SGGEmbStep(net, 2, seed=seeds[0], device=DEV, h=200, w=320, n_boxes=6, n_pairs=5, use_graph=graph)
    assert step.capture(warmup=1) == graph, step.graph_error
    losses = []
    if graph:
        for sd in seeds[1:]:
            step.reseed(sd)
            losses.append(float(step()))
        losses.append(float(step.flush()))
    else:
        for k, sd in enumerate(seeds):
            if k:
                step.reseed(sd)
            losses.append(float(step()))
    torch.cuda.synchronize()
    w = net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()
    w6 = net.vrd.fc6.fc.weight.detach()[:64].cpu().numpy().copy()
    step.opt.unfuse()
    return losses, w, w6


l0, w0, v0 = run(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
peak = float(np.abs(w0).max())
hist = []
for rep in range(N):
    l, w, v = run(True)
    d = np.abs(w.astype(np.float64) - w0)
    r = float(d.max() / peak)
    hist.append(r)
    if r > 5e-7:
        rows = np.nonzero(d.max(axis=1) > 5e-7 * peak)[0]
        cols = np.nonzero(d.max(axis=0) > 5e-7 * peak)[0]
        print("run %d: max-abs rel %.3e, L2 rel %.3e, loss dev %.2e; deviating filter rows %d of %d %s, columns %d of %d" % (
            rep, r, float(np.linalg.norm(d) / np.linalg.norm(w0)), max(abs(a - b) / abs(a) for a, b in zip(l0, l)),
            rows.size, w0.shape[0], rows[:8].tolist(), cols.size, w0.shape[1]))
print("%d captured runs vs the eager run: max-abs rel deviations sorted: %s" % (N, " ".join("%.1e" % x for x in sorted(hist))))
