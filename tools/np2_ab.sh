set -e
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest tests/test_gpu_x2.py tests/test_net_handle.py -x -q -m gpu > gpurun_out/np2_tests.log 2>&1
echo "== NP2 on ==" > gpurun_out/np2_ab.log
timeout -k 10 120 python tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype fp16 --x2 2 --wgrad 0 --iters 20 >> gpurun_out/np2_ab.log 2>&1
timeout -k 10 120 python tools/bench_predict2d.py fp16x2 >> gpurun_out/np2_ab.log 2>&1
echo "== NP2 off (IUNET_X2_NP2=0) ==" >> gpurun_out/np2_ab.log
IUNET_X2_NP2=0 timeout -k 10 120 python tools/bench_conv.py --dim 2 --size 512 --n 8 --dtype fp16 --x2 2 --wgrad 0 --iters 20 >> gpurun_out/np2_ab.log 2>&1
IUNET_X2_NP2=0 timeout -k 10 120 python tools/bench_predict2d.py fp16x2 >> gpurun_out/np2_ab.log 2>&1
tail -3 gpurun_out/np2_tests.log
