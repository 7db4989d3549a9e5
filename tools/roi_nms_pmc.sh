#!/bin/bash
# rocprofv3 evidence for the HBM-side kernels (ROIAlign / ROIPool / NMS / proposal layer): kernel trace + FETCH_SIZE +
# WRITE_SIZE as three separate passes per case (the counters do not fit one pass; never combined with a trace domain).
#   tools/roi_nms_pmc.sh [out_dir] -> <out_dir>/{b1,b4}/{trace,fetch,write} and profiles/r02_roi_nms_pmc.json
set -e
out=${1:-gpurun_out/roi_pmc}
export TMPDIR=/tmp
rm -rf $out; mkdir -p $out
for c in b1 b4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$c/trace -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$c/fetch -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$c/write -o t -- python3 tools/roi_nms_pmc_one.py $c > $out/$c.write.log 2>&1
done
python3 tools/roi_nms_pmc_summary.py $out profiles/r02_roi_nms_pmc.json
