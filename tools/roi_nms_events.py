#!/usr/bin/env python3
"""bench.py's ``also.roi_nms`` leg on its own (the same cases, the same event protocol): one line per case.

    python tools/roi_nms_events.py [substring of the case names]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

out = bench.run_roi_nms(torch.device("cuda:0"))
for k, v in out.items():
    if isinstance(v, dict) and (len(sys.argv) < 2 or sys.argv[1] in k):
        print("%-26s %8.1f us  %7.1f GB/s  frac %.3f" % (k, v["avg_launch_us"], v["achieved"], v["frac"]))
