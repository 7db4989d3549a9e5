#!/usr/bin/env python3
"""Summarise tools/roi_nms_pmc.sh: per case of bench.py's also.roi_nms -- launches, summed kernel time per launch (kernel trace),
HBM bytes per launch from the PMC passes, achieved GB/s against 8 TB/s; the kernels of a case are listed one by one.

FETCH_SIZE / WRITE_SIZE are KB (TCC_EA0 request counters).  Calibration of /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (16 B per lane) -> doubled here; WRITE_SIZE is exact
for 16-B-per-lane streaming stores and for float atomics (one dword per lane); other access widths are uncalibrated, so
`hbm_bytes_corrected` is an estimate for kernels that read 4 B per lane (flagged)."""
import collections, csv, glob, json, os, re, sys

out, dest = sys.argv[1], sys.argv[2]
HBM_PEAK = 8000.0
NARROW = ("roi_align_bwd_kernel", "roi_pool_bwd_kernel", "nms_mask_kernel", "nms_scan_pipelined_kernel")
OURS = ("roi_", "nms_", "rpn_", "sort_", "gather_dets", "write_rois", "bbox_overlaps", "fill", "Fill", "memset")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return re.split(r"[<(]", name, 1)[0]


res = {"command": "tools/roi_nms_pmc.sh: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate runs, one "
                  "process per case) -- python3 tools/roi_nms_pmc_one.py CASE", "hbm_peak_gbs": HBM_PEAK,
       "note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "cases": {}}
for log in sorted(glob.glob("%s/*.trace.log" % out)):
    case = os.path.basename(log)[:-len(".trace.log")]
    meta = {}
    for line in open(log):
        if line.startswith("{"):
            meta = json.loads(line)
    if not meta:
        continue
    N, alg = meta["launches"], meta["algorithmic_bytes"]
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
    kernels, tot_us, tot_hbm = {}, 0.0, 0.0
    for k, (n, ns) in sorted(dur.items(), key=lambda x: -x[1][1]):
        if n < N:                                             # set-up kernels of the case (randn, copies) run once or twice
            continue
        fe, wr = cnt["FETCH_SIZE"].get(k, [0, 0.0]), cnt["WRITE_SIZE"].get(k, [0, 0.0])
        per_launch = n / float(N)                             # launches of this kernel per launch of the op
        us = ns / 1e3 / N
        hbm = (2.0 * fe[1] + wr[1]) * 1024.0 / N
        kernels[k] = {"launches_per_op": per_launch, "us_per_op": us, "fetch_bytes_raw_per_op": fe[1] * 1024.0 / N,
                      "write_bytes_per_op": wr[1] * 1024.0 / N, "hbm_bytes_corrected_per_op": hbm,
                      "fetch_correction_uncalibrated": k in NARROW}
        tot_us += us
        tot_hbm += hbm
    if not kernels:
        continue
    res["cases"][case] = {"launches": N, "avg_us": tot_us, "algorithmic_bytes": alg, "algorithmic_gbs": alg / tot_us / 1e3,
                          "algorithmic_frac_of_8TBs": alg / tot_us / 1e3 / HBM_PEAK, "hbm_bytes_corrected": tot_hbm,
                          "hbm_gbs": tot_hbm / tot_us / 1e3, "traffic_over_algorithmic": tot_hbm / alg, "kernels": kernels}
json.dump(res, open(dest, "w"), indent=1)
for case, r in res["cases"].items():
    print("%-26s %8.1f us  algorithmic %6.2f MB -> %7.1f GB/s (%.3f of 8 TB/s)  HBM %7.2f MB (%.2fx)  [%s]" % (
        case, r["avg_us"], r["algorithmic_bytes"] / 1e6, r["algorithmic_gbs"], r["algorithmic_frac_of_8TBs"],
        r["hbm_bytes_corrected"] / 1e6, r["traffic_over_algorithmic"],
        ", ".join("%s %.1f us" % (k, v["us_per_op"]) for k, v in r["kernels"].items())))
