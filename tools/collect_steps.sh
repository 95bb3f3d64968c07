set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/steps; mkdir -p $OUT; RND=r05
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tmp_tr; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_tr -o tr -- python3 $R/tools/bench_train3d.py 10 > $OUT/train3d.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_tr/tr_kernel_trace.csv 13 40 > $OUT/${RND}_train_step_by_kernel.txt
python3 $R/tools/step_timeline.py $OUT/tmp_tr/tr_kernel_trace.csv 134 2 > $OUT/${RND}_train3d_step_timeline.txt; rm -rf $OUT/tmp_tr
rm -rf $OUT/tmp_t2; rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_t2 -o t2 -- python3 $R/tools/bench_train2d.py 8 > $OUT/train2d.log 2>&1
python3 $R/tools/step_profile.py $OUT/tmp_t2/t2_kernel_trace.csv 128 40 > $OUT/${RND}_train2d_step_by_kernel.txt
python3 $R/tools/step_timeline.py $OUT/tmp_t2/t2_kernel_trace.csv 141 2 > $OUT/${RND}_train2d_step_timeline.txt; rm -rf $OUT/tmp_t2
{ python3 $R/tools/bench_train3d.py 10; python3 $R/tools/bench_train3d.py 10 gn; } 2>/dev/null | grep -v amdgpu > $OUT/${RND}_train3d_batchnorm_groupnorm.txt
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/${RND}_bench_c3.json.log 2>/dev/null
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 > $OUT/${RND}_bench_c2.json.log 2>/dev/null
tail -3 $OUT/${RND}_train3d_step_timeline.txt; tail -2 $OUT/${RND}_train2d_step_timeline.txt; cat $OUT/${RND}_train3d_batchnorm_groupnorm.txt
