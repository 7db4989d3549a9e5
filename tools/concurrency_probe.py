#!/usr/bin/env python3
"""How much of the configs[1] step's time is a lack of independent work?  Replays K independent copies of the step graph
(own net, own streams) side by side and reports the time per step of the ensemble: if K = 2 runs much faster per step than
K = 1, a deeper pipeline inside ONE step (more backbone branches in flight) would pay.

usage: concurrency_probe.py [K ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = torch.device("cuda:0")
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                 "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
ks = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
steps, streams = [], []
for i in range(max(ks)):
    net = train.build_sgg_net(101, device=DEV)
    st = train.SGGEmbStep(net, 2, seed=1 + i, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert st.capture(warmup=2), st.graph_error
    torch.cuda.synchronize()
    steps.append(st)
    streams.append(s)
N = 40
for k in ks:
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            for st, s in zip(steps[:k], streams[:k]):
                with torch.cuda.stream(s):
                    st()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
    print("K=%d  %.3f ms per round, %.3f ms per step" % (k, dt, dt / k), flush=True)
