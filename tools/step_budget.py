#!/usr/bin/env python3
"""Per-step budget of the REPLAYED step graph from one `rocprofv3 --kernel-trace` csv of `bench.py --no-also --no-cpu-baseline`
(the timed form: one HIP graph per step, head of batch k beside the backbone branches of batch k+1).

One steady-state period = the window between two launches of an anchor kernel that occurs once per step.  Kernels are put
into kinds by name (and, for the fc6 forward, by its K = 50176 duration); per kind: launches, summed kernel time, and the
time of the period during which kernels of that kind are the ONLY ones running ("exclusive": what the step would lose if
the kind were free) or run beside others ("shared").  Per hardware queue (= graph branch): busy time by kind and idle
time -- the longest queue is the period's critical path.

usage: step_budget.py kernel_trace.csv [anchor substring = bce_rows_fwd] [periods_from_end = 3]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "bce_rows_fwd"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
marks = [r[0] for r in rows if anchor in r[3]]
t0, t1 = marks[-back - 1], marks[-back]
win = [r for r in rows if t0 <= r[0] < t1]
T = (t1 - t0) / 1e6


def kind(name, us):
    n = name.replace("(anonymous namespace)::", "")
    if "conv_wgrad2_f32" in n and re.search(r"true(, (true|false))?>", n.split("(")[0]):
        return "fused wgrad + SGD update (fc6 / fc7)"
    if "conv_wgrad" in n:
        return "head filter / data gradients (wgrad kernels)"
    if "conv_gemm_f32" in n or "conv_igemm_f32" in n:
        if us > 250.0:
            return "fc6 forward (K = 50176)"
        return "pointwise / plain GEMM (conv_gemm_f32)" if "conv_gemm_f32" in n else "implicit GEMM (stem, strided 3x3, Winograd planes, head convs)"
    if "wino" in n:
        return "Winograd transforms"
    if "roi_" in n:
        return "ROI pool"
    if "sgd_" in n or "adam" in n:
        return "optimizer (multi-tensor)"
    if any(t in n for t in ("epilogue", "bce_", "l2norm", "pair_gather", "maxpool", "weight_dgrad", "conv_epilogue")):
        return "head / backbone glue kernels of the library"
    return "aten / runtime kernels (copies, fills, elementwise)"


ks = [(s, e, q, kind(n, (e - s) / 1e3)) for s, e, q, n in win]
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, q, k in ks:
    by[k][0] += 1
    by[k][1] += (e - s) / 1e6
# exclusive / shared time per kind: sweep over the boundaries
pts = sorted({t0, t1} | {s for s, e, q, k in ks} | {min(e, t1) for s, e, q, k in ks})
excl, shared, idle = collections.Counter(), collections.Counter(), 0.0
active = sorted(ks)
for a, b in zip(pts, pts[1:]):
    live = {k for s, e, q, k in active if s <= a and e >= b}
    n_live = sum(1 for s, e, q, k in active if s <= a and e >= b)
    d = (b - a) / 1e6
    if not live:
        idle += d
    elif len(live) == 1 and n_live >= 1:
        excl[next(iter(live))] += d
    else:
        for k in live:
            shared[k] += d
print("period %.3f ms, %d kernels, %d queues; no kernel running: %.3f ms" % (T, len(win), len({q for s, e, q, k in ks}), idle))
print("%-66s %8s %10s %10s %10s" % ("kind", "launches", "sum ms", "only ms", "beside ms"))
for k, (n, ms) in sorted(by.items(), key=lambda x: -x[1][1]):
    print("%-66s %8d %10.3f %10.3f %10.3f" % (k, n, ms, excl[k], shared[k]))
print("%-66s %8d %10.3f" % ("total kernel time (overlapped branches add up beyond the period)", len(ks), sum(v[1] for v in by.values())))
print()
qs = collections.defaultdict(lambda: collections.Counter())
span = {}
for s, e, q, k in ks:
    qs[q][k] += (e - s) / 1e6
    span[q] = (min(span.get(q, (s, e))[0], s), max(span.get(q, (s, e))[1], e))
for q in sorted(qs, key=lambda q: -sum(qs[q].values())):
    busy = sum(qs[q].values())
    print("queue %d: busy %.3f ms, spans %.3f ms (%.3f .. %.3f)" % (q, busy, (span[q][1] - span[q][0]) / 1e6, (span[q][0] - t0) / 1e6, (span[q][1] - t0) / 1e6))
    for k, ms in qs[q].most_common():
        print("    %-62s %8.3f" % (k, ms))
