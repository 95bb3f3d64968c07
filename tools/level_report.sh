#!/bin/bash
# Per-level MFMA utilisation and HBM traffic of the 3x3x3 conv (forward) with rocprofv3 PMC counters.
# Run on the GPU box from the repo root: bash tools/level_report.sh ; then python tools/level_report.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/levels
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in 0:32:32 0:64:32 1:32:64 1:64:64 1:128:64 2:64:128 2:128:128 2:256:128 3:128:256 3:256:256; do
  tag=$(echo $shape | tr : _)
  python3 $R/tools/bench_conv.py --only $shape --wgrad 0 2>/dev/null > $OUT/time_$tag.txt
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma_$tag -- python3 $R/tools/bench_conv.py --only $shape --iters 2 --wgrad 0 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/bench_conv.py --only $shape --iters 2 --wgrad 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$tag -- python3 $R/tools/bench_conv.py --only $shape --iters 2 --wgrad 0 > /dev/null 2>&1
  echo done $shape
done
