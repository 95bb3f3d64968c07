"""C5 (BASELINE.json configs[4]): 3-D U-Net, 5 levels, base 64, 1 -> 4 classes, 128^3 chunks, bf16 activations.
Inference with e4m3 weights (per-output-channel power-of-two scales) and one bf16 training step."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet.unet import UNet
from interactive_unet.train_engine import TrainEngine

def flops_per_voxel(levels=5, base=64, cin=1, ncls=4, taps=27):
    ch = [base * 2 ** l for l in range(levels)]
    f = 0.0
    for l in range(levels):
        s = 1.0 / 8 ** l
        f += s * 2 * taps * ((cin if l == 0 else ch[l - 1]) * ch[l] + ch[l] * ch[l])
    for l in range(levels - 2, -1, -1):
        s = 1.0 / 8 ** l
        f += s * (2 * ch[l + 1] * ch[l] + 2 * taps * (2 * ch[l] * ch[l] + ch[l] * ch[l]))
    return f + 2 * ch[0] * ncls

S, N = 128, 1
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=4, dim=3, levels=5, base=64, act_dtype='bf16', pretrained=False, weight_dtype='fp8_e4m3')
m.reset_parameters(seed=0)
m = m.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(4)
x = torch.randint(1, 255, (N, 1, S, S, S), dtype=torch.uint8, device='cuda', generator=g)
eng = m.engine('eval')
probs = torch.empty((N, 4, S, S, S), device='cuda')
run = lambda: eng.infer(x, (S ** 3, S ** 3, S * S, S, 1), N, S, S, S, probs=probs)
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): run()
torch.cuda.synchronize(); ms = (time.time() - t0) / 5 * 1e3
fpv = flops_per_voxel()
print(f'C5 forward (e4m3 weights, bf16 activations), one 128^3 chunk: {ms:.2f} ms = {N * S ** 3 / ms / 1e3:.0f} Mvox/s, '
      f'{fpv * N * S ** 3 / ms / 1e9:.0f} TFLOP/s ({fpv:.0f} FLOP/voxel)')
te = TrainEngine(m.train(), lr=1e-4, loss_kind='mcc_ce')
lab = (x // 64).squeeze(1)
y = torch.stack([(lab == c) for c in range(4)], 1).half()
for _ in range(2): te.train_step(x, y, None, sync=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): te.train_step(x, y, None, sync=False)
torch.cuda.synchronize(); ms = (time.time() - t0) / 3 * 1e3
print(f'C5 training step (bf16), one 128^3 chunk: {ms:.2f} ms = {N * S ** 3 / ms / 1e3:.0f} Mvox/s, {3 * fpv * N * S ** 3 / ms / 1e9:.0f} TFLOP/s (3x fwd)')
