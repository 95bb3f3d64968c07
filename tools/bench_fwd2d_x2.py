"""2-D forward of the default prediction mode (x2m) at a batch / slice size:   python tools/bench_fwd2d_x2.py [N=8] [size=512]
(A/B switches of the library apply: IUNET_X2M_FIRST=2 runs encoder stage 0 as one launch, IUNET_X2M_POOL=0 keeps the pool launches)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'interactive-unet_amd'))
import torch
from interactive_unet.engine_x2 import EngineX2
from interactive_unet.unet import param_shapes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dim, shape = 2, (S, S)
g = torch.Generator().manual_seed(0)
p = {}
for n, s in param_shapes(dim).items():
    p[n] = (torch.rand(s, generator=g) + 0.5) if n.endswith('running_var') else torch.randn(s, generator=g) * 0.05
p = {k: v.cuda() for k, v in p.items()}
x = torch.randint(1, 255, (N, 1) + shape, dtype=torch.uint8, device='cuda')
probs = torch.empty((N, 2) + shape, device='cuda')
e = EngineX2(dim=dim)
e.load_eval(p)
vox = S * S
run = lambda: e.infer(x, (vox, vox, vox, S, 1), N, 1, S, S, probs=probs)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): run()
torch.cuda.synchronize(); print(f'2-D {N} x {S}^2 forward mixed={e.mixed}: {(time.time() - t0) / 20 * 1e3:.3f} ms')
