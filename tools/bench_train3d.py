"""The training leg of the C3 workload alone (3-D U-Net, bf16, 2 x 128^3 chunks per step): N timed steps after warm-up -- the thing to
put under rocprofv3 when working on the training kernels (tools/step_profile.py reads the trace).   python tools/bench_train3d.py [steps] [c3|c5]
(c5: 5 levels, base 64, 4 classes, one 128^3 chunk per step; gn: the C3 net with GroupNorm(8) instead of BatchNorm)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet.unet import UNet
from interactive_unet.train_engine import TrainEngine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
c5 = len(sys.argv) > 2 and sys.argv[2] == 'c5'
gn = len(sys.argv) > 2 and sys.argv[2] == 'gn'
NB, NC = (1, 4) if c5 else (2, 2)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = (UNet(lr=1e-4, num_classes=4, dim=3, levels=5, base=64, act_dtype='bf16', pretrained=False) if c5
         else UNet(lr=1e-4, num_classes=2, dim=3, act_dtype='bf16', pretrained=False, **({'norm': 'group'} if gn else {}))).cuda()
m.reset_parameters(seed=0)
te = TrainEngine(m, lr=1e-4, loss_kind='mcc_ce')
g = torch.Generator(device='cuda').manual_seed(1)
X = torch.randint(1, 255, (NB, 1, 128, 128, 128), dtype=torch.uint8, device='cuda', generator=g)
y = torch.cat([(X // (256 // NC)) == c for c in range(NC)], 1).to(torch.float16)
w = (torch.rand((NB, 1, 128, 128, 128), device='cuda', generator=g) > 0.1).to(torch.float16).expand(NB, NC, 128, 128, 128).contiguous()
y = y * w
for _ in range(3):
    te.train_step(X, y, w, sync=False)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    te.train_step(X, y, w, sync=False)
torch.cuda.synchronize()
print(f'train step {NB} x 128^3 bf16{" (C5 net)" if c5 else " (GroupNorm)" if gn else ""}: {(time.time() - t0) / steps * 1e3:.3f} ms')
