import os, sys, time, warnings
sys.path.insert(0,'interactive-unet_amd')
import torch
from interactive_unet.unet import UNet
S, C = 128, 2
g = torch.Generator(device='cuda').manual_seed(0)
blk = torch.randint(1, 255, (S, S, S), dtype=torch.uint8, device='cuda', generator=g)
out = torch.zeros((S, S, S, C), device='cuda')
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m = UNet(num_classes=2, dim=2, pretrained=False)
m.reset_parameters(seed=0); m = m.cuda().eval()
eng = m.engine('eval')
sb = (S*S, S, 1); so = (S*S*C, S*C, C)
rc = {0:(1,2),1:(0,2),2:(0,1)}
for axis in (0,1,2):
    r,c = rc[axis]
    f = lambda: eng.infer(blk.reshape(-1), (sb[axis],0,0,sb[r],sb[c]), S, 1, S, S, probs=out.reshape(-1), out_strides=(so[axis],1,0,so[r],so[c]), accumulate=True, divisor=1.0)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(f'axis {axis}: {(time.time()-t0)/20*1e3:.3f} ms per 128 slices ({eng.form})', flush=True)
