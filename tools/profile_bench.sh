#!/bin/bash
# The rocprofv3 evidence behind bench.py's roofline block; run on the GPU box from the repo root.
#   tools/profile_bench.sh [steps] [warmup]   -> gpurun_out/prof_bench/{trace,fetch,write} + gpurun_out/${TAG}_* (copy to profiles/)
set -e
steps=${1:-6}; warm=${2:-2}; TAG=${3:-r06}
export TMPDIR=/tmp
out=gpurun_out/prof_bench
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-also --no-graph --data resident > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o bench -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-also --no-graph --data resident > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o bench -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-also --no-graph --data resident > $out/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/mfma -o bench -- python3 bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-also --no-graph --data resident > $out/mfma.log 2>&1
# every eager step of the run: capture warm-up (2) + W + K + the 3 event-timed steps of the roofline block, see bench.py
python3 tools/pmc_summary.py $out $((steps + warm + 2 + 3)) gpurun_out/${TAG}

# configs[2]: kernel summary + the same two PMC passes -> gpurun_out/${TAG}_instance_styled_*
out=gpurun_out/prof_isd
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o isd -- python3 bench.py --config instance_styled --no-graph --steps 4 --warmup 1 --no-cpu-baseline > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o isd -- python3 bench.py --config instance_styled --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o isd -- python3 bench.py --config instance_styled --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/mfma -o isd -- python3 bench.py --config instance_styled --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $out/mfma.log 2>&1
# eager steps of the PMC runs: 2 warm-ups + W + K + the 2 event-timed steps of the roofline block
python3 tools/pmc_summary.py $out 7 gpurun_out/${TAG}_instance_styled
mv gpurun_out/${TAG}_instance_styled_bench_kernel_stats.csv gpurun_out/${TAG}_instance_styled_kernel_stats.csv
