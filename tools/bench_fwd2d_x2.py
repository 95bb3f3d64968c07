import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'interactive-unet_amd'))
import torch
from interactive_unet.engine_x2 import EngineX2
from interactive_unet.unet import param_shapes
dim, N, shape = 2, 8, (512, 512)
g = torch.Generator().manual_seed(0)
p = {}
for n, s in param_shapes(dim).items():
    p[n] = (torch.rand(s, generator=g) + 0.5) if n.endswith('running_var') else torch.randn(s, generator=g) * 0.05
p = {k: v.cuda() for k, v in p.items()}
x = torch.randint(1, 255, (N, 1) + shape, dtype=torch.uint8, device='cuda')
probs = torch.empty((N, 2) + shape, device='cuda')
e = EngineX2(dim=dim)
e.load_eval(p)
vox = 512 * 512
run = lambda: e.infer(x, (vox, vox, vox, 512, 1), N, 1, 512, 512, probs=probs)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): run()
torch.cuda.synchronize(); print(f'2-D 8 x 512^2 forward mixed={e.mixed}: {(time.time() - t0) / 10 * 1e3:.3f} ms')
