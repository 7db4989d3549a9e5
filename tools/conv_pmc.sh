#!/bin/bash
# SQ counters of the conv kernel on one layer3 shape; run on the GPU box from the repo root.
# usage: tools/conv_pmc.sh c2 out_dir
set -e
which=${1:-c2}; out=${2:-gpurun_out/pmc_$which}
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/set$i -o pmc -- python3 tools/conv_pmc_one.py $which > $out.set$i.log 2>&1 || { tail -5 $out.set$i.log; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-28s launches %3d  per launch %16.0f" % (k, n, v / n))
PY
