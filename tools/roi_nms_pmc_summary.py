#!/usr/bin/env python3
"""Summarise tools/roi_nms_pmc.sh: per case and cache state of bench.py's ROI_NMS_CASES -- kernel time per op (kernel trace), HBM
bytes per op from the PMC passes, achieved GB/s against 8 TB/s; the kernels of a case are listed one by one.  Only the kernels
the case names are counted; per kernel the rows are taken in dispatch order, the first ``dropped`` ops are left out and the
rest must be exactly ``ops x launches_per_op`` rows (asserted: a stray launch would show here).

FETCH_SIZE / WRITE_SIZE are KB (TCC_EA0 request counters).  Calibration of /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (16 B per lane) -> doubled here; WRITE_SIZE is exact
for 16-B-per-lane streaming stores and for float atomics (one dword per lane); other access widths are uncalibrated, so
`hbm_bytes_corrected` is an estimate for kernels that read 4 B per lane (flagged).  The counters sit on the L2's memory side:
a hit in the 256 MiB Infinity Cache counts like an HBM access, so "traffic" is fabric traffic in the warm state and HBM
traffic in the cold one."""
import collections, csv, glob, json, os, re, sys

out, dest = sys.argv[1], sys.argv[2]
HBM_PEAK = 8000.0
NARROW = ("roi_align_bwd_kernel", "roi_pool_bwd_kernel", "nms_mask_kernel", "nms_scan_pipelined_kernel", "nms_scan128_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return re.split(r"[<(]", name, 1)[0]


def rows_by_kernel(pattern, value):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            v = value(r)
            if v is not None:
                agg[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), v))
    return {k: [v for _, v in sorted(rows)] for k, rows in agg.items()}


res = {"command": "tools/roi_nms_pmc.sh: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate runs, one "
                  "process per case and state) -- python3 tools/roi_nms_pmc_one.py CASE warm|cold", "hbm_peak_gbs": HBM_PEAK,
       "note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "cases": {}}
for log in sorted(glob.glob("%s/*.trace.log" % out)):
    tag = os.path.basename(log)[:-len(".trace.log")]
    meta = {}
    for line in open(log):
        if line.startswith("{"):
            meta = json.loads(line)
    if not meta:
        print("no result line in", log)
        continue
    case, state, N, drop, alg = meta["case"], meta["state"], meta["ops"], meta["dropped"], meta["algorithmic_bytes"]
    dur = rows_by_kernel("%s/%s/trace/**/*kernel_trace.csv" % (out, tag), lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = {}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        cnt[counter] = rows_by_kernel("%s/%s/%s/**/*counter_collection.csv" % (out, tag, sub),
                                      lambda r, c=counter: float(r["Counter_Value"]) if r["Counter_Name"] == c else None)
    kernels, tot_us, tot_hbm = {}, 0.0, 0.0
    for k, lpo in meta["kernels"].items():
        def kept(rows, what):
            assert len(rows) == (N + drop) * lpo, "%s %s: %d %s rows of %s, expected %d" % (case, state, len(rows), what, k, (N + drop) * lpo)
            return rows[drop * lpo:]
        ns = kept(dur.get(k, []), "trace")
        fe, wr = kept(cnt["FETCH_SIZE"].get(k, []), "FETCH_SIZE"), kept(cnt["WRITE_SIZE"].get(k, []), "WRITE_SIZE")
        us = sum(ns) / 1e3 / N
        hbm = (2.0 * sum(fe) + sum(wr)) * 1024.0 / N
        kernels[k] = {"launches_per_op": lpo, "us_per_op": us, "us_min": min(ns) / 1e3 * lpo, "us_max": max(ns) / 1e3 * lpo,
                      "fetch_bytes_raw_per_op": sum(fe) * 1024.0 / N, "write_bytes_per_op": sum(wr) * 1024.0 / N,
                      "hbm_bytes_corrected_per_op": hbm, "fetch_correction_uncalibrated": k in NARROW}
        tot_us += us
        tot_hbm += hbm
    res["cases"].setdefault(case, {"algorithmic_bytes": alg})[state] = {
        "ops": N, "dropped": drop, "sets": meta["sets"], "avg_us": tot_us, "algorithmic_gbs": alg / tot_us / 1e3,
        "algorithmic_frac_of_8TBs": alg / tot_us / 1e3 / HBM_PEAK, "hbm_bytes_corrected": tot_hbm, "hbm_gbs": tot_hbm / tot_us / 1e3,
        "traffic_over_algorithmic": tot_hbm / alg, "kernels": kernels}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2vsgg_amd import build as _build  # noqa: E402
res["source_digests"] = _build.source_digests(["roi_ops.hip", "rpn.hip"])      # bench.py quotes this record only for these kernels
json.dump(res, open(dest, "w"), indent=1)
for case, states in res["cases"].items():
    for state in ("warm", "cold"):
        r = states.get(state)
        if r:
            print("%-24s %-4s %8.2f us  algorithmic %6.2f MB -> %7.1f GB/s (%.3f of 8 TB/s)  traffic %7.2f MB (%.2fx)  [%s]" % (
                case, state, r["avg_us"], states["algorithmic_bytes"] / 1e6, r["algorithmic_gbs"], r["algorithmic_frac_of_8TBs"],
                r["hbm_bytes_corrected"] / 1e6, r["traffic_over_algorithmic"],
                ", ".join("%s %.2f us" % (k, v["us_per_op"]) for k, v in r["kernels"].items())))
