# A/B of the 2-D cross-pair step (conv3_v4.hip NP in 2-D): IUNET_X2_NP2=0 / IUNET_NO_COMPACT2D=1 switch back to the padded operator.
# bash tools/np2_ab.sh  (one GPU; writes gpurun_out/np2_*.log)
set -e
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_x2.py tests/test_net_handle.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/np2_tests.log 2>&1 || { tail -30 gpurun_out/np2_tests.log; exit 1; }
tail -2 gpurun_out/np2_tests.log
for off in 0 1; do
  if [ $off = 1 ]; then export IUNET_NO_COMPACT2D=1; echo "== padded operator (IUNET_NO_COMPACT2D=1) =="; else echo "== cross-pair step =="; fi
  timeout -k 10 120 python tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype fp16 --wgrad 0 --iters 20
  timeout -k 10 120 python tools/bench_predict2d.py fp16
done > gpurun_out/np2_ab16.log 2>&1
