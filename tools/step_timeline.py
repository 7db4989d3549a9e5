#!/usr/bin/env python3
"""One steady-state replay of the step graph from a rocprofv3 --kernel-trace csv: every kernel of one period (between two
launches of the multi-tensor SGD kernel), by queue, with the idle time in front of it on its queue; per-queue busy time and
the time during which no kernel at all was running.

usage: step_timeline.py kernel_trace.csv [anchor_kernel_substring] [periods_from_end] [anchor_launches_per_step]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "sgd_momentum_multi"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
marks = [r[0] for r in rows if anchor in r[3]][::per_step]
t0, t1 = marks[-back - 1], marks[-back]
win = [r for r in rows if t0 <= r[0] < t1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return re.sub(r"\(.*", "", n)[:64]


print("period %.3f ms, %d kernels, %d queues" % ((t1 - t0) / 1e6, len(win), len({r[2] for r in win})))
last = {}
busy = collections.Counter()
for s, e, q, n in win:
    gap = (s - last[q]) / 1e3 if q in last else float("nan")
    last[q] = e
    busy[q] += e - s
    print("%9.1f %8.1f  q%-2d gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, short(n)))
ev = sorted([(s, 1) for s, e, q, n in win] + [(e, -1) for s, e, q, n in win])
depth, idle, prev = 0, 0, t0
hist = collections.Counter()
for t, d in ev:
    if t > t1:
        t = t1
    hist[depth] += t - prev
    prev = t
    depth += d
print("per-queue busy (ms):", {q: round(b / 1e6, 3) for q, b in busy.items()})
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
