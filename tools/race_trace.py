#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of tools/race_probe.py and reports which hardware queue every stream's
kernels ran on and whether consecutive replays of the head graph overlapped on the device.

usage: race_trace.py kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
has_stream = "Stream_Id" in rows[0]
R = []
for r in rows:
    R.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]),
              int(r["Stream_Id"]) if has_stream else -1, r["Kernel_Name"]))
R.sort()
t00 = R[0][0]
cnt = collections.Counter((q, s) for _, _, q, s, _ in R)
print("kernels per (queue, stream):", dict(cnt))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]


pool = [r for r in R if "roi_pool_fwd" in r[4]]
multi = [r for r in R if "sgd_momentum_multi" in r[4]]
mp = [r for r in R if "maxpool3x3s2" in r[4]]
print("roi_pool_fwd launches %d, sgd_momentum_multi %d, maxpool (backbone passes) %d" % (len(pool), len(multi), len(mp)))
print("queues of roi_pool_fwd:", collections.Counter((r[2], r[3]) for r in pool))
print("queues of sgd_multi   :", collections.Counter((r[2], r[3]) for r in multi))
print("queues of maxpool     :", collections.Counter((r[2], r[3]) for r in mp))
# head passes: first roi_pool of a pass = every second launch
firsts = pool[0::2]
viol = 0
for i in range(1, min(len(firsts), len(multi) + 1)):
    prev_end = multi[i - 1][1]
    if firsts[i][0] < prev_end:
        viol += 1
        print("OVERLAP: head pass %d starts %.1f us BEFORE the update of pass %d ends (queues %s -> %s)" % (
            i, (prev_end - firsts[i][0]) / 1e3, i - 1, (multi[i - 1][2], multi[i - 1][3]), (firsts[i][2], firsts[i][3])))
print("head-pass overlaps: %d of %d" % (viol, len(firsts) - 1))
# any two kernels of the same (queue, stream) overlapping in time?
last = {}
inq = 0
for s, e, q, st, n in R:
    k = (q, st)
    if k in last and s < last[k][1]:
        inq += 1
        if inq <= 10:
            print("same-queue overlap on %s: %s starts %.1f us before %s ends" % (k, short(n), (last[k][1] - s) / 1e3, short(last[k][4])))
    last[k] = (s, e, q, st, n)
print("same-(queue,stream) overlaps:", inq)
# cross-queue concurrency: time with kernels of >= 2 different queues in flight
ev = []
for s, e, q, st, n in R:
    ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort()
act = collections.Counter()
both = 0
prev = ev[0][0]
for t, d, q in ev:
    if sum(1 for v in act.values() if v > 0) >= 2:
        both += t - prev
    prev = t
    act[q] += d
print("time with >= 2 queues busy: %.3f ms of %.3f ms" % (both / 1e6, (R[-1][1] - t00) / 1e6))
