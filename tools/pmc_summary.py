#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_bench.sh into the files committed under profiles/.

  pmc_summary.py <out_dir> <steps_in_run> <profiles_prefix>

<out_dir>/fetch, <out_dir>/write: `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (csv); <out_dir>/trace: the
`--kernel-trace --stats` pass.  FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) and is doubled here."""
import collections, csv, glob, json, os, re, sys

out, steps, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return re.split(r"[<(]", name, 1)[0][:120]


per = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    per[counter] = {k: {"launches": n, "sum_kb": v} for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]}

# optional third PMC pass (tools/profile_bench.sh): SQ_VALU_MFMA_BUSY_CYCLES, summed over the chip's 1024 SIMDs by rocprofv3.
# Together with the trace pass's durations it gives the matrix-pipe utilisation from COUNTERS (not from a FLOP count):
#   util = busy cycles / (1024 SIMDs x launch duration x shader clock).  The clock is the one the chip holds under this
# load (2.06 GHz inside a lone conv_gemm_f32 launch, 2.22 GHz over the step: DESIGN.md 5.3 / 5.4); 2.1e9 is used here.
CLOCK_HZ = 2.1e9
mfma = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("%s/mfma/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        a = mfma[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("%s/trace/**/*kernel_stats.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        a = dur[short(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])


def mfma_block(kernel):
    n, busy = mfma.get(kernel, (0, 0.0))
    c, ns = dur.get(kernel, (0, 0.0))
    if not n or not c:
        return None
    per_launch, avg_ns = busy / n, ns / c
    return {"launches": n, "mfma_busy_cycles_per_launch": per_launch, "avg_duration_us": avg_ns / 1e3,
            "mfma_util_at_2.1GHz": per_launch / (1024.0 * avg_ns * 1e-9 * CLOCK_HZ),
            "tflops_from_busy_cycles_f32_16x16x4": per_launch / 32.0 * 2048.0 / (avg_ns * 1e-9) / 1e12}


def block(kernel):
    f = per["FETCH_SIZE"].get(kernel, {"launches": 0, "sum_kb": 0.0})
    w = per["WRITE_SIZE"].get(kernel, {"launches": 0, "sum_kb": 0.0})
    n = max(f["launches"], 1)
    return {"launches": f["launches"], "fetch_kb_raw": f["sum_kb"], "write_kb": w["sum_kb"],
            "hbm_bytes_per_launch_corrected": (2.0 * f["sum_kb"] + w["sum_kb"]) * 1024.0 / n,
            "hbm_bytes_per_step_corrected": (2.0 * f["sum_kb"] + w["sum_kb"]) * 1024.0 / max(steps, 1)}


summary = {
    "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps K --warmup W --no-cpu-baseline --no-also --no-graph",
    "steps_in_run": steps,
    "note": "FETCH_SIZE/WRITE_SIZE are KB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced streaming reads -> doubled.",
    "conv_gemm_f32": block("conv_gemm_f32"),
    "conv_igemm_f32": block("conv_igemm_f32"),
    "conv_wgrad2_f32": block("conv_wgrad2_f32"),
    "mfma_utilisation_from_counters": {k: mfma_block(k) for k in ("conv_gemm_f32", "conv_igemm_f32", "conv_wgrad2_f32")},
    "per_kernel": per,
}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2vsgg_amd import build as _build  # noqa: E402
summary["source_digests"] = _build.source_digests(["conv.hip", "winograd.hip"])     # bench.py quotes this record only for these kernels
json.dump(summary, open(prefix + "_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("conv_gemm_f32", "conv_igemm_f32")}))

# kernel stats of the trace pass -> csv (name, calls, total ns, avg ns, %)
rows = []
for f in glob.glob("%s/trace/**/*kernel_stats.csv" % out, recursive=True):
    rows = list(csv.reader(open(f)))
if rows:
    with open(prefix + "_bench_kernel_stats.csv", "w", newline="") as fo:
        csv.writer(fo).writerows(rows)
    print("kernel stats rows:", len(rows) - 1)
