#!/bin/bash
# Per-level MFMA utilisation and HBM traffic of the 3^d conv (forward; WGRAD=1: the weight gradient, tags wgrad_3d / wgrad_2d) with rocprofv3 PMC counters.
# Run on the GPU box from the repo root:   bash tools/level_report.sh [dim=3] [n=1] [size=128] [dtype=bf16] [tag=3d] [extra bench_conv flags, e.g. "--x2 2" with tag x2_3d]
# then   python tools/level_report.py [tag] [round]   merges gpurun_out/levels_<tag>/ into profiles/<round>_conv_levels_<tag>.md
DIM=${1:-3}; N=${2:-1}; SIZE=${3:-128}; DT=${4:-bf16}; TAG=${5:-3d}; EXTRA=${6:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/levels_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--dim $DIM --n $N --size $SIZE --dtype $DT --wgrad ${WGRAD:-0} $EXTRA"
SHAPES=${SHAPES:-"0:32:32 0:64:32 1:32:64 1:64:64 1:128:64 2:64:128 2:128:128 2:256:128 3:128:256 3:256:256"}      # C5 (tag f8_3d): SHAPES="0:64:64 0:128:64 1:64:128 ..." with EXTRA "--base 64 --levels 5 --f8 2"
for shape in $SHAPES; do
  tag=$(echo $shape | tr : _)
  python3 $R/tools/bench_conv.py --only $shape $ARGS --iters 30 2>/dev/null > $OUT/time_$tag.txt
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma_$tag -- python3 $R/tools/bench_conv.py --only $shape $ARGS --iters 2 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python3 $R/tools/bench_conv.py --only $shape $ARGS --iters 2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$tag -- python3 $R/tools/bench_conv.py --only $shape $ARGS --iters 2 > /dev/null 2>&1
  echo done $shape
done
