"""Ablation timing of the bf16 3-D Cout=32 conv (which phase costs what).  Profiling only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv
from bench_conv import timeit
S = 128
for cin in (32, 64):
    cout, taps, vox = 32, 27, S ** 3
    x = (torch.randn(cin * vox, device='cuda') * 0.5).to(torch.bfloat16)
    y = torch.empty(cout * vox, dtype=torch.bfloat16, device='cuda')
    w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, 0), dtype=torch.bfloat16, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    nv.call('iunet_pack_conv3', 1, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, 0, nv.stream())
    fl = 2.0 * taps * cin * cout * vox
    names = {0: 'full', 1: 'no weight loads', 2: 'no LDS frag reads', 4: 'no staging', 8: 'no stores', 3: 'no wloads+no LDS reads',
             7: 'MFMA + epilogue only', 15: 'MFMA only'}
    for e, name in names.items():
        f = lambda: nv.call('iunet_dbg_conv3_ablate', e, nv.ptr(x), nv.ptr(y), nv.ptr(wpk), nv.ptr(bias), 1, S, S, S, cin, cout, nv.stream())
        ms = timeit(f, iters=10)
        print(f'{cin}->32 mask {e:2d} {name:26s}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF/s-equivalent', flush=True)
