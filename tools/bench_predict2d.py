"""Latency of the interactive 2-D entry points (reference semantics): predict_slice on one 512^2 slice and the
2.5-D predict_block on one 128^3 block with the 2-D net.  Launch-bound or GPU-bound?"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import numpy as np, torch
from interactive_unet.unet import UNet
from interactive_unet import predict
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=2, dim=2, act_dtype=(sys.argv[1] if len(sys.argv) > 1 else 'fp16'), pretrained=False).cuda().eval()      # fp16 | bf16 | fp16x2 | fp32
rng = np.random.default_rng(0)
img = rng.integers(1, 255, (512, 512), dtype=np.uint8)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print('predict_slice 512^2 (host in, host out): %.3f ms' % t(lambda: predict.predict_slice(img, model=m)))
eng = m.engine('eval')
xd = torch.tensor(img).cuda(); probs = torch.empty(1, 2, 512, 512, device='cuda')
print('  device-only forward 512^2: %.3f ms' % t(lambda: eng.infer(xd, (512*512, 512*512, 512*512, 512, 1), 1, 1, 512, 512, probs=probs)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.infer(xd, (512*512, 512*512, 512*512, 512, 1), 1, 1, 512, 512, probs=probs)
e1.record(); torch.cuda.synchronize()
print('  GPU time per forward (events): %.3f ms' % (e0.elapsed_time(e1) / 20))
blk = torch.tensor(rng.integers(1, 255, (128, 128, 128), dtype=np.uint8)).cuda()
out = torch.empty(128, 128, 128, 2, device='cuda')
ms = t(lambda: predict.predict_block_device(m, blk, out, 2, None, (0, 1, 2)), n=10)
print('predict_block 2.5-D 128^3 (3 axes x 128 slices, on device): %.3f ms = %.1f Mvox/s' % (ms, 128**3 / ms / 1e3))
x8 = torch.tensor(rng.integers(1, 255, (8, 1, 512, 512), dtype=np.uint8)).cuda(); p8 = torch.empty(8, 2, 512, 512, device='cuda')
ms = t(lambda: eng.infer(x8, (512*512, 512*512, 512*512, 512, 1), 8, 1, 512, 512, probs=p8))
print('forward batch 8 x 512^2 (C2): %.3f ms = %.1f Mvox/s, %.0f TFLOP/s' % (ms, 8*512*512/ms/1e3, 280256*8*512*512/ms/1e9))
