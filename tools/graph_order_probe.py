#!/usr/bin/env python3
"""Which ingredient makes HIP-graph replays on the LEGACY DEFAULT stream go wrong under ROCm 7.2's packet-capture replay path
(DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default)?  Runs the configs[1] step (train.SGGEmbStep, one fork/join graph per
step) back to back without host synchronisation and compares the loss trajectory with a run that synchronises after
every step.

usage: [DEBUG_CLR_GRAPH_PACKET_CAPTURE=0|1] [I2V_SPLIT_ATOMICS=1] graph_order_probe.py MODE[,MODE..] [reps]
  MODE: {def|own}[_noarena][_seq]    def = replays on the legacy default stream, own = on a torch.cuda.Stream()
                                     noarena = no pre-zeroed arena: every atomically accumulated output is cleared by a
                                               hipMemsetAsync of its own, i.e. a MEMSET NODE in the graph
                                     seq = sequential graph (no second branch)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")       # this probe studies the default path unless told otherwise
import torch  # noqa: E402

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

train.REDIRECT_DEFAULT_STREAM = False      # replay where the caller stands, the default stream included
DEV = torch.device("cuda:0")
N = 23


def run(mode, sync=False):
    own = mode.startswith("own")
    net = train.build_sgg_net(101, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV, zero_arena="noarena" not in mode, overlap="seq" not in mode)
    prev = torch.cuda.current_stream()
    if own:
        s = torch.cuda.Stream()
        s.wait_stream(prev)
        torch.cuda.set_stream(s)
    try:
        assert step.capture(warmup=2), step.graph_error
        trace = torch.zeros(N, device=DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            trace[i].copy_(step())
            if sync or i == 2:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
        return trace.tolist(), net.vrd.fc7.fc.weight.detach().double().abs().sum().item(), dt
    finally:
        torch.cuda.set_stream(prev)
        step.opt.unfuse()


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["def"]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                     "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s I2V_SPLIT_ATOMICS=%s" % (os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"),
                                                                     os.environ.get("I2V_SPLIT_ATOMICS")), flush=True)
    ref = None
    for mode in modes:
        if ref is None:
            ref = run("own" + mode[3:], sync=True)
            print("REF final %.7f fc7 %.6f" % (ref[0][-1], ref[1]), flush=True)
        for r in range(reps):
            tr, w, dt = run(mode)
            dev = [abs(a - b) if a == a else float("inf") for a, b in zip(tr, ref[0])]
            first = next((i for i, d in enumerate(dev) if d > 2e-6), None)
            print("RUN %-16s rep %d  %.3f ms/step  final %.7f  fc7 %.6f  first_dev_step %s  max_dev %.3g" % (
                mode, r, dt, tr[-1], w, first, max(dev)), flush=True)


if __name__ == "__main__":
    main()
