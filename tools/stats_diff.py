#!/usr/bin/env python3
"""Per-kernel total time of two rocprofv3 --stats runs of the same command side by side (old vs new build on one box).
usage: stats_diff.py old_kernel_stats.csv new_kernel_stats.csv"""
import csv
import re
import sys


def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
        n = re.sub(r"\(.*", "", n)
        n = re.sub(r", (false|true)(, (false|true))?>$", ">", n)       # old / new builds differ in trailing template flags
        c, t = d.get(n, (0, 0))
        d[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
print("%-70s %6s %10s | %6s %10s | %9s" % ("kernel", "calls", "old ms", "calls", "new ms", "delta ms"))
for k in rows[:25]:
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    print("%-70s %6d %10.3f | %6d %10.3f | %+9.3f" % (k[:70], ca, ta / 1e6, cb, tb / 1e6, (tb - ta) / 1e6))
print("total old %.3f ms, new %.3f ms" % (sum(v[1] for v in a.values()) / 1e6, sum(v[1] for v in b.values()) / 1e6))
