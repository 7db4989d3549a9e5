#!/usr/bin/env python3
"""Summarise tools/roi_nms_pmc.sh: per kernel and case, launches, average duration (kernel trace), HBM bytes per launch
from the PMC passes and the achieved GB/s against 8 TB/s.

FETCH_SIZE / WRITE_SIZE are KB (TCC_EA0 request counters).  Calibration of /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (16 B per lane) -> doubled here; WRITE_SIZE is exact
for 16-B-per-lane streaming stores and for float atomics (one dword per lane); other access widths are uncalibrated, so
`hbm_bytes_corrected` is an estimate for kernels that read 4 B per lane (flagged)."""
import collections, csv, glob, json, re, sys

out, dest = sys.argv[1], sys.argv[2]
HBM_PEAK = 8000.0
NARROW = ("roi_align_bwd_kernel", "roi_pool_bwd_kernel", "nms_mask_kernel", "nms_scan_pipelined_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return re.split(r"[<(]", name, 1)[0]


res = {"command": "tools/roi_nms_pmc.sh: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate runs) -- "
                  "python3 tools/roi_nms_pmc_one.py CASE", "hbm_peak_gbs": HBM_PEAK,
       "note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "cases": {}}
for case in ("b1", "b4"):
    alg = {}
    for line in open("%s/%s.trace.log" % (out, case)):
        if line.startswith("{"):
            alg = json.loads(line)["algorithmic_bytes"]
    dur = collections.defaultdict(lambda: [0, 0])
    for f in glob.glob("%s/%s/trace/**/*kernel_trace.csv" % (out, case), recursive=True):
        for r in csv.DictReader(open(f)):
            d = dur[short(r["Kernel_Name"])]
            d[0] += 1
            d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt = {}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob("%s/%s/%s/**/*counter_collection.csv" % (out, case, sub), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    a = agg[short(r["Kernel_Name"])]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
        cnt[counter] = agg
    rows = {}
    for k, (n, ns) in sorted(dur.items(), key=lambda x: -x[1][1]):
        if not any(t in k for t in ("roi_", "nms_", "rpn_", "sort_", "gather_dets", "write_rois", "bbox_overlaps")):
            continue
        fe, wr = cnt["FETCH_SIZE"].get(k, [0, 0.0]), cnt["WRITE_SIZE"].get(k, [0, 0.0])
        us = ns / n / 1e3
        fetch_b = fe[1] * 1024.0 / max(fe[0], 1)
        write_b = wr[1] * 1024.0 / max(wr[0], 1)
        hbm = 2.0 * fetch_b + write_b
        row = {"launches": n, "avg_us": us, "fetch_bytes_raw": fetch_b, "write_bytes": write_b, "hbm_bytes_corrected": hbm,
               "hbm_gbs": hbm / us / 1e3, "hbm_frac_of_8TBs": hbm / us / 1e3 / HBM_PEAK,
               "fetch_correction_uncalibrated": k in NARROW}
        if k in alg:
            row["algorithmic_bytes"] = alg[k]
            row["algorithmic_gbs"] = alg[k] / us / 1e3
            row["algorithmic_frac_of_8TBs"] = alg[k] / us / 1e3 / HBM_PEAK
            row["traffic_over_algorithmic"] = hbm / alg[k] if alg[k] else None
        rows[k] = row
    res["cases"][case] = rows
json.dump(res, open(dest, "w"), indent=1)
for case, rows in res["cases"].items():
    print("case", case)
    for k, r in rows.items():
        print("  %-32s x%-3d %8.1f us  HBM %7.2f MB -> %7.1f GB/s (%.2f of 8 TB/s)%s" % (
            k, r["launches"], r["avg_us"], r["hbm_bytes_corrected"] / 1e6, r["hbm_gbs"], r["hbm_frac_of_8TBs"],
            "  algorithmic %.2f MB -> %.1f GB/s" % (r["algorithmic_bytes"] / 1e6, r["algorithmic_gbs"]) if "algorithmic_bytes" in r else ""))
