#!/usr/bin/env python3
"""Which stream handle plays which role (round-3 review, item 2).  Prints, for one process that builds the step objects the
way the test suite does: torch's pooled handles in dealing order, the handle of torch.cuda.graph's capture stream, and the
registry's table (ops.role_stream) with the order in which the step objects asked for streams -- with torch.cuda.Stream() every
request drew the NEXT pooled handle, so request k aliased request k - 32.  A top-level program (never a child of a GPU process).

    python tools/stream_handles.py > profiles/r04_stream_handles.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import i2vsgg_amd  # noqa: E402,F401
from i2vsgg_amd import eval as ev, ops, train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = torch.device("cuda:0")
c.cfg_from_file(c.default_cfg_file("res101"))
pool = [torch.cuda.Stream(DEV).cuda_stream for _ in range(34)]
print("torch pool, dealing order (34 draws):", " ".join("%#x" % h for h in pool))
print("distinct pooled handles: %d; draw 32 == draw 0: %s" % (len(set(pool)), pool[32] == pool[0]))
print("default stream %#x" % torch.cuda.default_stream(DEV).cuda_stream)

net = train.build_sgg_net(layers=50, seed=5, device=DEV)
net.vrd.dropout = False
for k in range(3):
    step = train.SGGEmbStep(net, 2, seed=3, device=DEV, h=200, w=320, n_boxes=6, n_pairs=5)
    assert step.capture(warmup=1), step.graph_error
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    step.opt.unfuse()
    step = None
cap = getattr(torch.cuda.graph, "default_capture_stream", None)
print("torch.cuda.graph capture stream: %s (pooled: %s)" % ("%#x" % cap.cuda_stream if cap is not None else None,
                                                          cap is not None and cap.cuda_stream in pool))
print("registry:")
for (dev, role, prio), h in sorted(ops.stream_table().items(), key=lambda kv: str(kv[0])):
    print("  device %d  %-14s priority %d  %#x  pooled: %s" % (dev, role, prio, h, h in pool))
hs = list(ops.stream_table().values())
print("registry handles distinct: %s; any in torch's pool: %s" % (len(set(hs)) == len(hs), bool(set(hs) & set(pool))))
print("requests in order (%d): %s" % (len(ops.STREAM_REQUESTS), " ".join(str(r) for r in ops.STREAM_REQUESTS)))
print("with torch.cuda.Stream() these requests would have drawn pooled handles %d..%d: request k and request k+32 alias"
      % (34, 34 + len(ops.STREAM_REQUESTS) - 1))
