"""The 3x3x3 stage convs of the C3 network in the two split-precision forms, layer by layer: fp16x2 (three 16-bit MFMAs per product,
conv3_v4.hip SPL) against x2m (main term 16-bit + both cross terms on the K = 128 fp8 instruction, conv3_x2m.hip).
`python tools/bench_x2m.py [N] [size]` -> us per launch, algorithmic TF/s, ratio."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
from interactive_unet import _native as nv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
DIM = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # 2: the 2-D stage convs (`python tools/bench_x2m.py 8 512 2`)
TAPS = 3 ** DIM
A = 64.0
LAYERS = [(0, 32, 32), (0, 64, 32), (1, 32, 64), (1, 64, 64), (1, 128, 64), (2, 64, 128), (2, 128, 128), (2, 256, 128), (3, 128, 256), (3, 256, 256)]


def timeit(run, iters=20):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for lvl, ci, co in LAYERS:
    d = S >> lvl
    vox = d ** DIM
    dd = (d, d, d) if DIM == 3 else (1, d, d)
    dev = 'cuda'
    w = (torch.randn((co, ci) + (3,) * DIM, device=dev) * (2.0 / (ci * TAPS)) ** 0.5).contiguous()
    s = nv.stream()
    # fp16x2
    wv = torch.empty(3 * ci * co * TAPS, device=dev)
    osc, b = torch.empty(co, device=dev), torch.empty(co, device=dev)
    nv.call('iunet_x2_prep', nv.ptr(w), nv.ptr(wv), nv.ptr(osc), nv.ptr(b), None, None, None, None, None, 1e-5, A, A, co, ci, TAPS, 0, 16 if DIM == 3 else 32, s)
    pm = nv.lib().iunet_x2_pack_mode(DIM)
    wpk = torch.empty(nv.pack_conv3_elems(co, 3 * ci, TAPS, pm), dtype=torch.float16, device=dev)
    nv.call('iunet_pack_conv3', 0, nv.ptr(wv), None, nv.ptr(wpk), co, 3 * ci, TAPS, pm, s)
    xs = (torch.rand(N * 2 * ci * vox, device=dev) * 100).to(torch.float16)
    y = torch.empty(N * 2 * co * vox, dtype=torch.float16, device=dev)
    t_x2 = timeit(lambda: nv.call('iunet_x2_conv3_fwd', DIM, nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(y), 2 * co * vox, co // 8, nv.ptr(wpk),
                                  nv.ptr(osc), nv.ptr(b), N, *dd, ci, co, 2, s))
    # x2m
    whi = torch.empty(co * ci * TAPS, device=dev)
    w8 = torch.zeros(nv.lib().iunet_x2m_w8_bytes_nd(DIM, co, ci), dtype=torch.uint8, device=dev)
    nv.call('iunet_x2m_prep_nd', DIM, nv.ptr(w), nv.ptr(whi), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), None, None, None, None, 1e-5, A, A, co, ci, s)
    pm16 = 2 if DIM == 3 else 6
    w16 = torch.empty(nv.pack_conv3_elems(co, ci, TAPS, pm16), dtype=torch.float16, device=dev)
    nv.call('iunet_pack_conv3', 0, nv.ptr(whi), None, nv.ptr(w16), co, ci, TAPS, pm16, s)
    x8 = torch.empty(N * 2 * ci * vox, dtype=torch.uint8, device=dev)
    nv.call('iunet_x2m_make8', nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(x8), 2 * ci * vox, ci, N, *dd, s)
    y8 = torch.empty(N * 2 * co * vox, dtype=torch.uint8, device=dev)
    res = {}
    for name, ylo, yy8 in (('hi+lo+m8', co // 8, y8), ('hi+m8', -1, y8), ('hi+lo', co // 8, None)):
        res[name] = timeit(lambda: nv.call('iunet_x2m_conv_fwd', DIM, nv.ptr(xs), 2 * ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), 2 * co * vox, ylo,
                                           nv.ptr(yy8), 2 * co * vox, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), N, *dd, ci, co, 2, None, s))
    pool_txt = ''
    if ci == co and lvl < 3:                # an encoder stage's second conv: the stage's max-pool in its epilogue against its own launch
        pv = vox // 2 ** DIM
        pd = tuple(max(v // 2, 1) for v in dd)
        py = torch.empty(N * co * pv, dtype=torch.float16, device=dev)
        py8 = torch.empty(N * 2 * co * pv, dtype=torch.uint8, device=dev)
        t_f = timeit(lambda: nv.call('iunet_x2m_conv_pool_fwd', DIM, nv.ptr(xs), 2 * ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), 2 * co * vox, -1,
                                     nv.ptr(y8), 2 * co * vox, nv.ptr(py), co * pv, nv.ptr(py8), 2 * co * pv, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b),
                                     N, *dd, ci, co, 2, None, s))
        t_p = timeit(lambda: nv.call('iunet_x2m_maxpool_fwd', DIM, nv.ptr(y), 2 * co * vox, nv.ptr(y8), 2 * co * vox, nv.ptr(py), co * pv, nv.ptr(py8),
                                     2 * co * pv, co, N, *pd, s))
        pool_txt = f' | conv+pool in one launch {t_f:8.1f} us against {res["hi+m8"]:.1f} + {t_p:.1f} us'
    fl = 2.0 * TAPS * ci * co * vox * N
    print(f'L{lvl} {ci:3d}->{co:3d} @ {N} x {d}^{DIM}: fp16x2 {t_x2:8.1f} us ({fl / t_x2 / 1e6:6.1f} TF/s alg) | x2m ' +
          ' '.join(f'{k} {v:8.1f} us' for k, v in res.items()) + f' | x2m/fp16x2 = {res["hi+lo+m8"] / t_x2:.3f} ({fl / res["hi+m8"] / 1e6:6.1f} TF/s alg)' + pool_txt, flush=True)
