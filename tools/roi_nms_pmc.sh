#!/bin/bash
# rocprofv3 evidence for the HBM-side kernels behind bench.py's roofline_hbm / also.roi_nms (ROIAlignAvg fwd / bwd, ROIPool over
# packed maps, NMS), warm and cold: kernel trace + FETCH_SIZE + WRITE_SIZE as three separate passes per case and state (the
# counters do not fit one pass; never combined with a trace domain).  One process per case, state and pass, so that a kernel's
# rows belong to exactly one case.
#   tools/roi_nms_pmc.sh [out_dir] [dest.json] [cases...] -> <out_dir>/<case>.<state>/{trace,fetch,write} and profiles/r06_roi_nms_pmc.json
set -e
out=${1:-gpurun_out/roi_pmc}
dest=${2:-profiles/r06_roi_nms_pmc.json}
shift 2 2>/dev/null || true
export TMPDIR=/tmp
rm -rf $out; mkdir -p $out
cases=${@:-$(python3 -c "import bench; print(' '.join(bench.ROI_NMS_CASES))")}
for c in $cases; do
  for st in warm cold; do
    d=$out/$c.$st
    rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python3 tools/roi_nms_pmc_one.py $c $st > $d.trace.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -o t -- python3 tools/roi_nms_pmc_one.py $c $st > $d.fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -o t -- python3 tools/roi_nms_pmc_one.py $c $st > $d.write.log 2>&1
    echo "done $c $st"
  done
done
python3 tools/roi_nms_pmc_summary.py $out $dest
