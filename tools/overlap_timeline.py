#!/usr/bin/env python3
"""What the two-stream step costs the backbone: compares a rocprofv3 kernel trace of the overlapped schedule with one
of the one-stream schedule (I2V_OVERLAP=0) and reports, per head kernel, how fast the backbone stream advanced while
that kernel was running (1.0 = as fast as alone).

usage: overlap_timeline.py overlapped_kernel_trace.csv sequential_kernel_trace.csv"""
import collections
import csv
import re
import sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Queue_Id"]), r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[2])
    return rows


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[(]", name)[0][:60]


def passes(rows):
    """Backbone passes: [stem conv, maxpool, ..., last kernel before the next stem conv / before a head kernel]."""
    idx = [i for i, r in enumerate(rows) if "maxpool3x3s2" in r[1]]
    return idx


ov, sq = load(sys.argv[1]), load(sys.argv[2])
# overlapped: the backbone queue is the one that holds the maxpool
bbq = collections.Counter(r[0] for r in ov if "maxpool3x3s2" in r[1]).most_common(1)[0][0]
bb = [r for r in ov if r[0] == bbq]
hd = [r for r in ov if r[0] != bbq]
mp = [i for i, r in enumerate(bb) if "maxpool3x3s2" in r[1]]
# steady state: the last 4 graph passes on the backbone queue
starts = [i - 1 for i in mp]
n_bb = starts[-1] - starts[-2]
print("backbone queue %d: %d kernels per pass, %d passes on it" % (bbq, n_bb, len(starts)))
# sequential trace: solo duration of the k-th kernel of a pass (median over the last passes)
smp = [i for i, r in enumerate(sq) if "maxpool3x3s2" in r[1]]
solo = []
for k in range(n_bb):
    d = sorted(sq[i - 1 + k][3] - sq[i - 1 + k][2] for i in smp[-4:] if i - 1 + k < len(sq))
    solo.append(d[len(d) // 2])
    assert short(sq[smp[-1] - 1 + k][1]) == short(bb[starts[-2] + k][1]), (k, sq[smp[-1] - 1 + k][1], bb[starts[-2] + k][1])
print("backbone pass alone: %.3f ms busy" % (sum(solo) / 1e6))
# overlapped steady-state passes (not the last one: the run ends under it)
use = starts[-4:-1]
tot_ov = 0
segs = []           # (start, end, solo_ns) of every backbone kernel in the analysed passes
for s in use:
    for k in range(n_bb):
        r = bb[s + k]
        segs.append((r[2], r[3], solo[k]))
        tot_ov += r[3] - r[2]
span = [(bb[s][2], bb[s + n_bb - 1][3]) for s in use]
print("backbone pass overlapped: %.3f ms busy, %.3f ms first-start to last-end (median)" % (
    tot_ov / len(use) / 1e6, sorted(e - b for b, e in span)[len(span) // 2] / 1e6))


def progress(a, b):
    """solo-nanoseconds of backbone work done inside [a,b) (a kernel advances uniformly over its own duration)."""
    p = 0.0
    for s, e, so in segs:
        lo, hi = max(a, s), min(b, e)
        if hi > lo:
            p += so * (hi - lo) / (e - s)
    return p


t0, t1 = span[0][0], span[-1][1]
agg = collections.defaultdict(lambda: [0, 0, 0.0])
for q, name, s, e in hd:
    if s < t0 or e > t1:
        continue
    k = short(name)
    agg[k][0] += 1
    agg[k][1] += e - s
    agg[k][2] += progress(s, e)
# head-stream idle time inside the window
hs = sorted((s, e) for q, n, s, e in hd if s >= t0 and e <= t1)
idle_prog, idle_t, cur = 0.0, 0, t0
for s, e in hs:
    if s > cur:
        idle_t += s - cur
        idle_prog += progress(cur, s)
    cur = max(cur, e)
print("%-62s %6s %9s %9s" % ("head kernel (overlapped run, per step)", "calls", "ms", "bb rate"))
n = len(use)
for k, (c, t, p) in sorted(agg.items(), key=lambda x: -x[1][1])[:22]:
    print("%-62s %6.1f %9.3f %9.2f" % (k, c / n, t / n / 1e6, p / t if t else 0))
print("%-62s %6s %9.3f %9.2f" % ("(head stream idle)", "", idle_t / n / 1e6, idle_prog / idle_t if idle_t else 0))
sq_names = collections.defaultdict(list)
for q, name, s, e in sq[smp[-4] - 1:]:
    sq_names[short(name)].append(e - s)
print("\nsolo durations of the head kernels (one-stream run):")
for k, (c, t, p) in sorted(agg.items(), key=lambda x: -x[1][1])[:10]:
    d = sq_names.get(k)
    if d:
        print("  %-60s avg %8.1f us alone, %8.1f us overlapped" % (k, sum(d) / len(d) / 1e3, t / c / 1e3))
