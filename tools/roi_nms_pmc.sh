#!/bin/bash
# rocprofv3 evidence for the HBM-side kernels behind bench.py's also.roi_nms (ROIAlignAvg fwd / bwd, ROIPool over packed maps,
# NMS): kernel trace + FETCH_SIZE + WRITE_SIZE as three separate passes per case (the counters do not fit one pass; never
# combined with a trace domain).  One process per case and pass, so that a kernel's rows belong to exactly one case.
#   tools/roi_nms_pmc.sh [out_dir] -> <out_dir>/<case>/{trace,fetch,write} and profiles/r04_roi_nms_pmc.json
set -e
out=${1:-gpurun_out/roi_pmc}
export TMPDIR=/tmp
rm -rf $out; mkdir -p $out
cases="roi_align_avg_fwd_1x32 roi_align_avg_bwd_1x32 roi_align_avg_fwd_4x32 roi_align_avg_bwd_4x32 roi_pool_geom_fwd_2x64 nms_12000_to_2000 nms_6000_to_300"
for c in $cases; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$c/trace -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$c/fetch -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$c/write -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.write.log 2>&1
  echo "done $c"
done
python3 tools/roi_nms_pmc_summary.py $out profiles/r04_roi_nms_pmc.json
