#!/usr/bin/env python3
"""The shader clock the chip holds while the configs[1] step graph replays back to back (two in-stream stamps of the
shader-clock counter and the 100 MHz counter, 200 steps apart), next to the clock of an idle-ish stretch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import _lib, train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = torch.device("cuda:0")
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                 "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
net = train.build_sgg_net(101, device=DEV)
step = train.SGGEmbStep(net, 2, seed=1, device=DEV)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
assert step.capture(warmup=2), step.graph_error
st = torch.zeros(8, 2, dtype=torch.int64, device=DEV)


def stamp(i):
    _lib.lib.i2v_debug_clock_stamp(st[i].data_ptr(), torch.cuda.current_stream().cuda_stream)


for _ in range(100):
    step()
stamp(0)
for _ in range(200):
    step()
stamp(1)
torch.cuda.synchronize()
time.sleep(0.5)
stamp(2)
torch.cuda.synchronize()
time.sleep(0.2)
stamp(3)
torch.cuda.synchronize()
v = st.cpu().double()
clk = lambda a, b: (v[b, 0] - v[a, 0]) / (v[b, 1] - v[a, 1]) * 100e6 / 1e9
print("200 steps: %.3f ms per step, shader clock %.3f GHz" % ((v[1, 1] - v[0, 1]) / 100e6 / 200 * 1e3, clk(0, 1)))
print("idle 0.2 s: shader clock counter rate %.3f GHz" % clk(2, 3))
