#!/usr/bin/env python3
"""Which chain of the relation step's graph is the long pole?  Clock stamps written INSIDE the captured graph (a one-wave kernel
that stores s_memrealtime, 100 MHz: i2v_debug_clock_stamp; HIP has no timing for captured events) at the first fork on the
capturing stream, at the end of every forked branch and on the capturing stream right before the join, read after replays:
when did the head (capturing stream) and each frame's backbone branch finish, relative to the fork.
  tools/branch_ends.py [HxW ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

from i2vsgg_amd import ops, train  # noqa: E402
from i2vsgg_amd._lib import lib  # noqa: E402

DEV = torch.device("cuda:0")
marks = {}
buf = torch.zeros((16, 2), dtype=torch.int64, device=DEV)


class Stamp:
    n = 0

    def __init__(self):
        self.i = Stamp.n
        Stamp.n += 1

    def record(self, st):
        assert lib.i2v_debug_clock_stamp(buf[self.i].data_ptr(), st.cuda_stream) == 0


def _event(enable_timing=True):
    return Stamp()


real_join, real_enter = ops.join, ops.branch.__enter__


def enter(self):
    if "start" not in marks and torch.cuda.is_current_stream_capturing():
        e = _event()
        e.record(self.origin)
        marks["start"] = e
    return real_enter(self)


def join(origin, *streams):
    if torch.cuda.is_current_stream_capturing() and "start" in marks and "main" not in marks:
        e = _event()
        e.record(origin)
        marks["main"] = e
        for i, st in enumerate(streams):
            e = _event()
            e.record(st)
            marks["branch%d" % i] = e
    real_join(origin, *streams)
    if torch.cuda.is_current_stream_capturing() and "joined" not in marks and "main" in marks:
        e = _event()
        e.record(origin)
        marks["joined"] = e


ops.branch.__enter__ = enter
ops.join = join
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(600, 1000)]
for h, w in sizes:
    marks.clear()
    Stamp.n = 0
    net = train.build_sgg_net(101, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV, h=h, w=w)
    assert step.capture(warmup=2), step.graph_error
    rows = []
    for i in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        if i >= 4:
            t = buf.cpu()[:, 1]
            rows.append([(int(t[marks[k].i]) - int(t[marks["start"].i])) * 1e-5 for k in sorted(marks) if k != "start"] + [e0.elapsed_time(e1)])
    names = [k for k in sorted(marks) if k != "start"] + ["whole replay"]
    med = [sorted(r[j] for r in rows)[len(rows) // 2] for j in range(len(names))]
    print("== %dx%d (one replay at a time, ms after the fork): " % (h, w) + ", ".join("%s %.3f" % (n, v) for n, v in zip(names, med)))
    step.opt.unfuse()
    del step, net
    torch.cuda.empty_cache()
