"""2-D training step (the reference UI's configuration: batch 8 x 512^2, fp16, MCC+CE): wall time per step vs GPU time."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import numpy as np, torch
from interactive_unet.unet import UNet
from interactive_unet.train_engine import TrainEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=2, dim=2, act_dtype='fp16', pretrained=False).cuda()
te = TrainEngine(m, lr=1e-4, loss_kind='mcc_ce')
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.randint(1, 255, (B, 1, 512, 512), dtype=torch.uint8, device='cuda', generator=g)
lab = X > 127
y = torch.cat([~lab, lab], 1).half()
w = torch.ones_like(y)
for _ in range(3): te.train_step(X, y, w, sync=False)
torch.cuda.synchronize(); t0 = time.time()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): te.train_step(X, y, w, sync=False)
e1.record(); torch.cuda.synchronize()
wall = (time.time() - t0) / n * 1e3
print(f'2-D train step, batch {B} x 512^2 fp16: wall {wall:.3f} ms/step, GPU (events) {e0.elapsed_time(e1) / n:.3f} ms/step, '
      f'{B * 512 * 512 / wall / 1e3:.1f} Mvox/s, {3 * 280256 * B * 512 * 512 / wall / 1e9:.0f} TFLOP/s (3x fwd)')
t0 = time.time()
for _ in range(n): te.train_step(X, y, w, sync=False)
host = (time.time() - t0) / n * 1e3
torch.cuda.synchronize()
print(f'host time to enqueue one step: {host:.3f} ms')
