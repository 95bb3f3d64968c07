"""The prediction forward of BASELINE config C5 alone (3-D, 5 levels, base 64, 4 classes, e4m3 weights + activations on the K = 128 fp8 matrix
instruction): N timed forwards of one 128^3 block after warm-up -- the thing to put under rocprofv3 (tools/step_profile.py reads the trace).
python tools/bench_c5_predict.py [iters]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet.unet import UNet
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=4, dim=3, levels=5, base=64, act_dtype='bf16', pretrained=False, weight_dtype='fp8_e4m3').cuda().eval()
m.reset_parameters(seed=0)
eng = m.engine('eval')
x = torch.randint(0, 256, (1, 1, 128, 128, 128), dtype=torch.uint8, device='cuda')
vox = 128 ** 3
cls = torch.empty((1, vox), dtype=torch.uint8, device='cuda')
for _ in range(3):
    eng.infer(x, (vox, vox, 128 * 128, 128, 1), 1, 128, 128, 128, cls=cls)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(iters):
    eng.infer(x, (vox, vox, 128 * 128, 128, 1), 1, 128, 128, 128, cls=cls)
torch.cuda.synchronize()
print(f'C5 forward 128^3, e4m3 planes = {eng.q_planes()}: {(time.time() - t0) / iters * 1e3:.3f} ms')
