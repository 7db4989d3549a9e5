#!/usr/bin/env python3
"""bench.py's ``also.roi_nms`` leg on its own (the same cases, both cache states, the same event protocol): one line per
case and state.

    python tools/roi_nms_events.py [substring of the case names]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

out = bench.run_roi_nms(torch.device("cuda:0"), only=sys.argv[1] if len(sys.argv) > 1 else None)
for k, v in out.items():
    if isinstance(v, dict):
        for st in ("warm", "cold"):
            r = v[st]
            print("%-26s %-4s events %8.2f us (frac %.3f)  rocprof %s us  sets %d" % (
                k, st, r["events_us"], r["frac_events"], "%8.2f" % r["rocprof_us"] if r["rocprof_us"] else "    n/a", r["sets"]))
