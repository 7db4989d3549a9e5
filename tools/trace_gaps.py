#!/usr/bin/env python3
"""Busy time vs gaps between kernels from a rocprofv3 --kernel-trace database (rocpd .db).
usage: trace_gaps.py results.db [n_last_kernels]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  positive gaps: n %d sum %.3f ms median %.2f us  overlapped pairs %d" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, len(pos), sum(pos) / 1e6, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0,
    sum(1 for g in gaps if g <= 0)))
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e in rows:
    k = re.sub(r"\(.*", "", name); k = re.sub(r"<.*", "", k)[:48]
    agg[k][0] += 1; agg[k][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-48s %6d launches %9.3f ms  avg %7.1f us" % (k, c, t / 1e6, t / c / 1e3))
short = lambda name: re.sub(r"<.*", "", re.sub(r"\(.*", "", name)).replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at::")[:40]
ctx = collections.defaultdict(lambda: [0, 0])
for i, g in enumerate(gaps):
    if g > 2000:
        k = short(rows[i][0]) + "  ->  " + short(rows[i + 1][0])
        ctx[k][0] += 1; ctx[k][1] += g
print("gaps > 2 us by (previous -> next kernel):")
for k, (c, t) in sorted(ctx.items(), key=lambda x: -x[1][1])[:25]:
    print("  %-86s %5d  %8.3f ms  avg %6.1f us" % (k, c, t / 1e6, t / c / 1e3))
