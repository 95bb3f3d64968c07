#!/bin/bash
# SQ counters of the 2-D 16-bit stage conv alone (three --pmc passes per layer, 4 counters each): what the consumer side of conv3_v4.hip's
# 2-D variants is busy with.   bash tools/pmc_conv2d.sh > profiles/r05_pmc_conv2d_lds.txt   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for layer in 0:32:32 0:64:32 1:32:64 1:128:64; do
  echo "== layer $layer @ 8 x 512^2 fp16 (level:Cin:Cout)"
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    rm -rf /tmp/pmc2d
    rocprofv3 --pmc $set --output-format csv -d /tmp/pmc2d -- python3 $R/tools/bench_conv.py --only $layer --dim 2 --size 512 --n 8 --dtype f16 --wgrad 0 --iters 3 > /dev/null 2>&1
    python3 $R/tools/pmc_kernel.py /tmp/pmc2d conv3_v4
  done
done
