"""The prediction leg of the C3 workload alone (one rank): tiled prediction of a 416 x 128 x 128 uint8 volume = 2 overlapping 128^3 blocks
(reflect-padded gather, batched forward + softmax, Gaussian blend, normalise / quantise) -- the thing to put under rocprofv3
(tools/step_profile.py reads the trace).   python tools/bench_predict3d.py [reps] [bf16|fp16x2]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet.unet import UNet
from interactive_unet import shard
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=2, dim=3, act_dtype='bf16', infer_dtype='fp16x2' if mode == 'fp16x2' else 'bf16', pretrained=False).cuda().eval()
m.reset_parameters(seed=0)
S, V = 128, (416, 128, 128)
vol = torch.randint(0, 256, V, dtype=torch.uint8, device='cuda')
ops = shard.NativeOps(m, 2, S)
for _ in range(3):
    shard.predict_volume_sharded(ops, vol, V, S, 0.25, group=None)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(reps):
    shard.predict_volume_sharded(ops, vol, V, S, 0.25, group=None)
torch.cuda.synchronize()
print(f'predict 416 x 128 x 128 ({mode}): {(time.time() - t0) / reps * 1e3:.3f} ms per volume')
