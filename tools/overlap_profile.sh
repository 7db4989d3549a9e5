#!/bin/bash
# kernel traces of the two-stream step and of the one-stream step -> tools/overlap_timeline.py
set -e
export TMPDIR=/tmp
out=gpurun_out/overlap
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/ov -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/ov.log 2>&1
I2V_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $out/sq -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/sq.log 2>&1
python3 tools/overlap_timeline.py $(find $out/ov -name "*kernel_trace.csv") $(find $out/sq -name "*kernel_trace.csv") > $out/timeline.txt 2>&1 || true
rm -f $(find $out -name "*kernel_trace.csv") 
tail -60 $out/timeline.txt
