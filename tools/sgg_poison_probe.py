#!/usr/bin/env python3
"""Does the captured configs[1] step read memory it never wrote?  The caching allocator's free blocks are filled with a
poison value (0, 1e30, NaN) before the step objects are built; the loss trajectory of 23 replays must not depend on it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = torch.device("cuda:0")
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                 "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])


def poison(val):
    junk = [torch.full((256 << 20,), val, device=DEV) for _ in range(12)]
    small = [torch.full((n,), val, device=DEV) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26) for _ in range(8)]
    torch.cuda.synchronize()
    del junk, small


def run():
    net = train.build_sgg_net(101, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(s)
    try:
        assert step.capture(warmup=2), step.graph_error
        tr = torch.zeros(23, device=DEV)
        for i in range(23):
            tr[i].copy_(step())
        torch.cuda.synchronize()
        return tr.tolist()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        step.opt.unfuse()


ref = None
for val in (0.0, 1e30, float("nan"), -1e30, float("nan")):
    poison(val)
    tr = run()
    if ref is None:
        ref = tr
    dev = max(abs(a - b) if a == a else float("inf") for a, b in zip(tr, ref))
    print("poison %-6s final %.7f  max deviation from the first run %.3g" % (val, tr[-1], dev), flush=True)
    torch.cuda.empty_cache()
