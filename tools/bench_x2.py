"""Time the split-precision mode (engine_x2.EngineX2) beside the fp32 parity mode and the 16-bit engines: one forward of the 3-D net
on a 128^3 chunk and of the 2-D net on 8 x 512^2 slices, plus the per-layer times of the 3-D stage convs.   python tools/bench_x2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet.engine import Engine
from interactive_unet.engine_f32 import EngineF32
from interactive_unet.engine_x2 import EngineX2
from interactive_unet.unet import param_shapes

def flops_per_voxel(dim, levels=4, base=32, cin=1, ncls=2):
    ch = [base * 2 ** l for l in range(levels)]; taps, f = 3 ** dim, 0.0
    for l in range(levels):
        f += 2 * taps * ((cin if l == 0 else ch[l - 1]) * ch[l] + ch[l] * ch[l]) / 2 ** (dim * l)
    for l in range(levels - 2, -1, -1):
        f += (2 * ch[l + 1] * ch[l] + 2 * taps * (2 * ch[l] * ch[l] + ch[l] * ch[l])) / 2 ** (dim * l)
    return f + 2 * ch[0] * ncls

which = sys.argv[1:] or ['x2', 'f32', 'bf16', 'fp16']
for dim, N, shape in ((3, 1, (128, 128, 128)), (3, 2, (128, 128, 128)), (2, 8, (512, 512))):
    g = torch.Generator().manual_seed(0)
    p = {}
    for n, s in param_shapes(dim).items():
        p[n] = (torch.rand(s, generator=g) + 0.5) if n.endswith('running_var') else torch.randn(s, generator=g) * 0.05
    p = {k: v.cuda() for k, v in p.items()}
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    x = torch.randint(1, 255, (N, 1) + shape, dtype=torch.uint8, device='cuda')
    probs = torch.empty((N, 2) + shape, device='cuda')
    mk = {'x2': lambda: EngineX2(dim=dim), 'f32': lambda: EngineF32(dim=dim), 'bf16': lambda: Engine(dim=dim, act_dtype=torch.bfloat16),
          'fp16': lambda: Engine(dim=dim, act_dtype=torch.float16)}
    for name in which:
        e = mk[name]()
        e.load_eval(p)
        run = lambda: e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, probs=probs)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): run()
        torch.cuda.synchronize(); ms = (time.time() - t0) / 10 * 1e3
        print(f'{dim}-D {N} x {shape} forward, {name}: {ms:.2f} ms = {N * vox / ms / 1e3:.0f} Mvox/s, {flops_per_voxel(dim) * N * vox / ms / 1e9:.1f} TFLOP/s (algorithmic)', flush=True)
        del e
        torch.cuda.empty_cache()
