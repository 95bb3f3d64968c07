#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run from the repo root through gpurun); the summaries land in
# gpurun_out/profiles_$RND/ and are copied into profiles/ (tracked) afterwards.   [PART=A|B] bash tools/collect_profiles.sh [r05]
RND=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$RND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <name> <program args...>: rocprofv3 --kernel-trace --stats, keep the kernel_stats.csv
  local name=$1; shift
  rm -rf $OUT/tmp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -- "$@" > $OUT/${name}.log 2>&1
  cp $(ls $OUT/tmp_$name/*/*_kernel_stats.csv | head -1) $OUT/${RND}_${name}_kernel_stats.csv
  cp $(ls $OUT/tmp_$name/*/*_kernel_trace.csv | head -1) $OUT/${name}_kernel_trace.csv
  rm -rf $OUT/tmp_$name
  echo "done stats $name"
}
pmc() {     # pmc <name> <counter> <program args...>
  local name=$1 ctr=$2; shift 2
  rm -rf $OUT/tmp_pmc
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/tmp_pmc -- "$@" > /dev/null 2>&1
  cp $(ls $OUT/tmp_pmc/*/*_counter_collection.csv | head -1) $OUT/${RND}_pmc_${ctr}_${name}.csv
  rm -rf $OUT/tmp_pmc
  echo "done pmc $name $ctr"
}
# PART=A: sections 1-3 (bench lines, rocprofv3 summaries, roofline kernels); PART=B: the rest; unset: everything (two calls fit gpurun's 20 minutes)
if [ "${PART:-A}" = "A" ] || [ -z "${PART+x}" ]; then
# 1. plain bench lines (unprofiled) of every workload
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/${RND}_bench_c3.json.log 2>/dev/null; echo "done bench c3"
python3 $R/bench.py --workload c4 --steps 3 --warmup 1 > $OUT/${RND}_bench_c4.json.log 2>/dev/null; echo "done bench c4"
python3 $R/bench.py --workload c5 --steps 10 --warmup 3 > $OUT/${RND}_bench_c5.json.log 2>/dev/null; echo "done bench c5"
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 > $OUT/${RND}_bench_c2.json.log 2>/dev/null; echo "done bench c2"
# 2. rocprofv3 summaries of the same commands (short runs)
stats bench_c3 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --c4-reps 0
stats bench_c5 python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline
stats bench_c2 python3 $R/bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline
# the launches of the roofline layer (dec0.conv1 forward) inside the profiled steps: what bench.py's in-situ `roofline` times
python3 $R/tools/roofline_launches.py $OUT/bench_c3_kernel_trace.csv 'conv3_v4_kernelIDF16bLi3ELb0ELb0ELb0ELb0ELb1ELb0E' 300 460 262144 1 > $OUT/${RND}_bench_c3_roofline_launches.txt      # 256 workgroups x 1024 threads: the bf16 compact-operator variant (8 loader waves)
python3 $R/tools/roofline_launches.py $OUT/bench_c5_kernel_trace.csv conv3_f8k_kernel 250 600 98304 2 > $OUT/${RND}_bench_c5_roofline_launches.txt      # 2 x 128 workgroups x 768 threads: the K = 128 kernel on e4m3 planes
python3 $R/tools/roofline_launches.py $OUT/bench_c2_kernel_trace.csv conv3_v4_kernel 85 140 196608 1 > $OUT/${RND}_bench_c2_roofline_launches.txt
# 3. the roofline kernels alone: 3 warm-up + 50 launches, the sequence bench.py times
stats roofline_c3 python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 50 --n 2
stats roofline_c3_1tile python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 50
stats roofline_c5 python3 $R/tools/bench_conv.py --only 0:128:64 --base 64 --levels 5 --wgrad 0 --f8 2 --iters 50      # e4m3 planes in and out, as the engine launches it
stats roofline_c2 python3 $R/tools/bench_conv.py --only 0:64:32 --dim 2 --size 512 --n 8 --dtype f16 --wgrad 0 --iters 50
fi
if [ "${PART:-B}" = "B" ] || [ -z "${PART+x}" ]; then
# the split-precision (fp16x2) conv of the same layer alone, and one whole split-precision forward per configuration
stats roofline_x2 python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 50 --n 2 --x2 2
stats roofline_x2m python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 50 --n 2 --x2m 2      # the predict leg's kernel: cross terms on the fp8 matrix cores
stats forward_x2 python3 $R/tools/bench_x2.py x2
python3 $R/tools/bench_conv.py --wgrad 0 --iters 30 --n 1 --x2 1 --x2m 1 > $OUT/${RND}_conv_layers_x2_3d.txt 2>/dev/null
python3 $R/tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype f16 --wgrad 0 --iters 30 --x2 1 --x2m 1 > $OUT/${RND}_conv_layers_x2_2d.txt 2>/dev/null
# x2m stage convs layer by layer incl. the pooled encoder convs (conv + pool in one launch against two), the 2-D first stage in one launch
python3 $R/tools/bench_x2m.py 2 128 3 > $OUT/${RND}_x2m_layers_3d.txt 2>/dev/null
python3 $R/tools/bench_x2m.py 8 512 2 > $OUT/${RND}_x2m_layers_2d.txt 2>/dev/null
{ python3 $R/tools/bench_first_stage.py; python3 $R/tools/bench_first_stage.py 128 128; python3 $R/tools/bench_first_stage.py 48 128; } 2>/dev/null | grep -v amdgpu > $OUT/${RND}_first_stage_2d_with_pool.txt      # (r04_first_stage_2d.txt: the same tool before it timed the pooled combinations)
# forward and weight gradient per stage shape: 2-D (the dy-reuse form against the first form) and 3-D at the C3 step's batch of 2
python3 $R/tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype fp16 --iters 20 > $OUT/${RND}_conv_wgrad_layers_2d.txt 2>/dev/null
IUNET_WGRAD2D_V1=1 python3 $R/tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype fp16 --iters 20 > $OUT/${RND}_conv_wgrad_layers_2d_first_form.txt 2>/dev/null
python3 $R/tools/bench_conv.py --dim 3 --size 128 --n 2 --dtype bf16 --iters 10 > $OUT/${RND}_conv_wgrad_layers_3d_n2.txt 2>/dev/null
# every C5 stage-conv shape: 16-bit kernel and the K = 128 fp8 kernel on e4m3 planes side by side
python3 $R/tools/bench_conv.py --base 64 --levels 5 --f8 2 --wgrad 0 --iters 30 > $OUT/${RND}_conv_layers_c5_f8.txt 2>/dev/null
# every transposed-conv shape of C3 / C5 / C2 alone
python3 $R/tools/bench_convT.py 30 > $OUT/${RND}_convT_layers.txt 2>/dev/null
# the training step and the C5 prediction forward by kernel
rm -rf $OUT/tmp_tr; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_tr -o tr -- python3 $R/tools/bench_train3d.py 10 > $OUT/train3d.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_tr/tr_kernel_trace.csv 13 40 > $OUT/${RND}_train_step_by_kernel.txt
python3 $R/tools/step_timeline.py $OUT/tmp_tr/tr_kernel_trace.csv 134 2 > $OUT/${RND}_train3d_step_timeline.txt; rm -rf $OUT/tmp_tr      # one step in launch order (134 launches)
rm -rf $OUT/tmp_c5; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_c5 -o c5 -- python3 $R/tools/bench_c5_predict.py 10 > $OUT/c5_predict.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_c5/c5_kernel_trace.csv 13 20 > $OUT/${RND}_c5_forward_by_kernel.txt; rm -rf $OUT/tmp_c5
rm -rf $OUT/tmp_t2; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_t2 -o t2 -- python3 $R/tools/bench_train2d.py 8 > $OUT/train2d.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_t2/t2_kernel_trace.csv 128 40 > $OUT/${RND}_train2d_step_by_kernel.txt
python3 $R/tools/step_timeline.py $OUT/tmp_t2/t2_kernel_trace.csv 141 2 > $OUT/${RND}_train2d_step_timeline.txt; rm -rf $OUT/tmp_t2      # 2 modes x (4 + 30 + 30) steps
rm -rf $OUT/tmp_pp; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_pp -o pp -- python3 $R/tools/bench_predict3d.py 10 fp16x2 > $OUT/predict3d.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_pp/pp_kernel_trace.csv 26 30 > $OUT/${RND}_predict_x2m_by_kernel.txt; rm -rf $OUT/tmp_pp      # 13 volumes of 4 blocks = 26 two-block forwards
python3 $R/tools/bench_train2d.py 1 8 2>/dev/null | grep -v amdgpu > $OUT/${RND}_train2d_latency.txt
{ python3 $R/tools/bench_train3d.py 10; python3 $R/tools/bench_train3d.py 10 gn; } 2>/dev/null | grep -v amdgpu > $OUT/${RND}_train3d_batchnorm_groupnorm.txt      # the C3 training step with BatchNorm and with GroupNorm(8)
python3 $R/tools/bench_predict_volumes.py 3 512 2>/dev/null | tail -1 > $OUT/${RND}_predict_volumes.txt; python3 $R/tools/bench_predict_volumes.py 2 1024 2>/dev/null | tail -1 >> $OUT/${RND}_predict_volumes.txt
python3 $R/tools/bench_latency.py 2>/dev/null | grep -v amdgpu > $OUT/${RND}_slice_latency.txt
python3 $R/tools/bench_2p5d.py 2>/dev/null | grep -v amdgpu > $OUT/${RND}_predict_2p5d_views.txt      # the 2.5-D block as one batch of views against three forwards
echo "done layer tables and step profiles"
# 4. HBM traffic of the roofline kernels (separate passes, as the guide prescribes)
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c3 $c python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 2
  pmc c5 $c python3 $R/tools/bench_conv.py --only 0:128:64 --base 64 --levels 5 --wgrad 0 --f8 2 --iters 2
  pmc c2 $c python3 $R/tools/bench_conv.py --only 0:64:32 --dim 2 --size 512 --n 8 --dtype f16 --wgrad 0 --iters 2
  pmc x2 $c python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 2 --x2 2
  pmc x2m $c python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 2 --x2m 2
done
python3 $R/tools/pmc_json.py $OUT $RND c3 conv3_v4_kernel 'dec0.conv1 64->32 @ 1 x 128^3 bf16' 402653184 1
python3 $R/tools/pmc_json.py $OUT $RND c5 conv3_f8k_kernel 'dec0.conv1 128->64 @ 1 x 128^3, e4m3 planes in and out, K = 128 fp8 MFMA' 402653184 1
python3 $R/tools/pmc_json.py $OUT $RND c2 conv3_v4_kernel 'dec0.conv1 64->32 @ 8 x 512^2 f16' 402653184 8
python3 $R/tools/pmc_json.py $OUT $RND x2 conv3_v4_kernel 'dec0.conv1 64->32 @ 1 x 128^3 fp16x2 (hi + lo planes in and out)' 805306368 1
python3 $R/tools/pmc_json.py $OUT $RND x2m conv3_x2m_kernel 'dec0.conv1 64->32 @ 1 x 128^3 x2m (hi + lo8 planes in and out: 3 bytes per element)' 603979776 1
# 5. per-level MFMA pipe busy + HBM bytes of the stage convs (north_star: "for the 3x3 conv at each resolution level"): the 16-bit kernels and
#    the split-precision kernel with its cross terms on the fp8 matrix cores, 3-D and 2-D
cd $R
bash tools/level_report.sh 3 1 128 bf16 3d > /dev/null 2>&1;               python3 tools/level_report.py 3d $RND
bash tools/level_report.sh 3 1 128 bf16 x2m_3d "--x2m 2" > /dev/null 2>&1; python3 tools/level_report.py x2m_3d $RND
bash tools/level_report.sh 2 8 512 f16 2d > /dev/null 2>&1;                python3 tools/level_report.py 2d $RND
bash tools/level_report.sh 2 8 512 f16 x2m_2d "--x2m 2" > /dev/null 2>&1;  python3 tools/level_report.py x2m_2d $RND
cp profiles/${RND}_conv_levels_*.md $OUT/ 2>/dev/null
for f in conv3_v4 conv3_x2m conv3_f8k conv3_wgrad_v2 conv2_wgrad_v2 conv3_wgrad pointwise split16 train_misc gn_precise train_pointwise; do python3 tools/regreport.py interactive-unet_amd/csrc/$f.hip; done > $OUT/${RND}_register_report.txt 2>&1
fi
ls -la $OUT
