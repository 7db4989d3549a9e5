#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of tools/race_probe.py and reports, per (hardware queue, stream), how often a
kernel STARTED before the previous kernel of the same stream had ENDED (a stream is in-order: any overlap beyond the
timestamp granularity is a lost dependency), with the largest cases by name.

usage: race_trace.py kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
has_stream = "Stream_Id" in rows[0]
R = []
for r in rows:
    R.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]),
              int(r["Stream_Id"]) if has_stream else -1, r["Kernel_Name"], int(r["Dispatch_Id"])))
R.sort(key=lambda r: r[5])          # dispatch order = submission order
if "Scratch_Size" in rows[0]:
    sc = collections.Counter()
    for r in rows:
        if int(r["Scratch_Size"]) > 0:
            sc[(int(r["Queue_Id"]), int(r["Stream_Id"]) if has_stream else -1, int(r["Scratch_Size"]),
                r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70])] += 1
    print("kernels that use scratch (private segment) memory: %d kinds" % len(sc))
    for k, v in sorted(sc.items()):
        print("   queue %d stream %d scratch %6d B x%-5d %s" % (k[0], k[1], k[2], v, k[3]))
cnt = collections.Counter((q, s) for _, _, q, s, _, _ in R)
print("kernels per (queue, stream):", dict(cnt))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "").split("(")[0][:44]


mp = collections.Counter((r[2], r[3]) for r in R if "maxpool3x3s2" in r[4])
pool = collections.Counter((r[2], r[3]) for r in R if "roi_pool_fwd" in r[4])
print("backbone passes by (queue, stream):", dict(mp))
print("head passes by (queue, stream)    :", {k: v // 2 for k, v in pool.items()})
last = {}
stats = collections.defaultdict(lambda: [0, 0, 0, 0, 0])       # n_pairs, >0.5us, >2us, >10us, >50us
worst = collections.defaultdict(list)
for s, e, q, st, n, d in R:
    k = (q, st)
    if k in last:
        ps, pe, pn = last[k]
        stats[k][0] += 1
        ov = pe - s
        if ov > 0:
            for i, th in enumerate((500, 2000, 10000, 50000)):
                if ov > th:
                    stats[k][i + 1] += 1
            if ov > 2000:
                worst[k].append((ov, short(n), short(pn)))
    if k not in last or e > last[k][1]:
        last[k] = (s, e, n)
print("%-10s %8s %10s %8s %8s %8s" % ("(q,stream)", "pairs", ">0.5us", ">2us", ">10us", ">50us"))
for k in sorted(stats):
    print("%-10s %8d %10d %8d %8d %8d" % ((str(k),) + tuple(stats[k])))
for k in sorted(worst):
    w = sorted(worst[k], reverse=True)[:8]
    print("largest overlaps on %s:" % (k,))
    for ov, n, pn in w:
        print("   %8.1f us: %-44s started before %-44s ended" % (ov / 1e3, n, pn))
