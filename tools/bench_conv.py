"""Micro-benchmark of the MFMA conv kernels per U-Net layer shape (HIP events on the launch stream).
    python tools/bench_conv.py [--dtype bf16] [--n 1] [--size 128] [--dim 3]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16'); ap.add_argument('--n', type=int, default=1)
    ap.add_argument('--size', type=int, default=128); ap.add_argument('--dim', type=int, default=3)
    ap.add_argument('--wgrad', type=int, default=1)
    ap.add_argument('--only', default='', help='e.g. 0:64:32 = level:cin:cout')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--layout', type=int, default=-1, help='-1 = library policy, 0 / 1 / 2 / 3 = force a kernel structure (3: compact operator, padding-free step)')
    ap.add_argument('--base', type=int, default=32, help='channels at level 0 (64: the C5 network)')
    ap.add_argument('--levels', type=int, default=4)
    ap.add_argument('--f8', type=int, default=0, help='1: also time the fp8 matrix-core kernel (conv3_f8.hip / conv3_f8k.hip) on each shape; 2: with e4m3 activation planes in and out (3-D, the engine\'s format between fp8 convs)')
    ap.add_argument('--x2', type=int, default=0, help='1: also time the split-precision conv (fp16x2: conv3_v4.hip SPL) on each shape; 2: only it')
    ap.add_argument('--x2m', type=int, default=0, help='1: also time the split-precision conv with its cross terms on the fp8 matrix cores (conv3_x2m.hip; '
                    'hi + lo8 planes in and out) on each shape; 2: only it')
    a = ap.parse_args()
    T = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    dt = nv.DTYPE_CODE[T]; nd = a.dim; taps = 3 ** nd
    b = a.base
    shapes = [(0, b, b), (0, 2 * b, b)]
    for l in range(1, a.levels):
        c = b << l
        shapes += [(l, c // 2, c), (l, c, c)] + ([(l, 2 * c, c)] if l < a.levels - 1 else [])
    if a.only:
        l, ci, co = [int(v) for v in a.only.split(':')]
        shapes = [(l, ci, co)]
    tot_f = tot_t = 0
    for lvl, cin, cout in shapes:
        S = a.size >> lvl
        D = S if nd == 3 else 1
        vox = D * S * S
        x = (torch.randn(a.n * cin * vox, device='cuda') * 0.5).to(T)
        y = torch.empty(a.n * cout * vox, dtype=T, device='cuda')
        w = torch.randn(cout, cin, *([3] * nd), device='cuda') * 0.05
        if a.layout >= 0:
            lay = a.layout
        elif nv.lib().iunet_conv3_compact_ok(nd, a.n, D, S, S, cin, cout, 0, 0):
            lay = 3                                    # what PackedConv.pick chooses for a plain launch (compact operator)
        else:
            lay = nv.lib().iunet_conv3_pick_layout(nd, a.n, D, S, S, cin, cout)
        pm = 6 if lay == 3 else 2 * (lay > 0)            # layout 3: the compact K16 order (pack mode bit 2)
        wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pm), dtype=T, device='cuda')
        bias = torch.zeros(cout, device='cuda')
        nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pm, nv.stream())
        f = lambda: nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wpk), nv.ptr(bias), None,
                            a.n, D, S, S, cin, cout, 2, lay, nv.stream())
        ms = timeit(f, iters=a.iters) if (a.x2 != 2 and a.x2m != 2) else float('nan')
        fl = 2.0 * taps * cin * cout * vox * a.n
        line = f'L{lvl} {cin:3d}->{cout:3d} @{S}^{nd} N={a.n} layout {lay}: fwd {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s'
        tot_f += fl; tot_t += ms
        if a.wgrad:
            dy = (torch.randn(a.n * cout * vox, device='cuda') * 0.5).to(T)
            nfl = nv.lib().iunet_conv3_wgrad_slab_floats(nd, a.n, D, S, S, cin, cout)
            slab = torch.empty(nfl, device='cuda'); dW = torch.empty(cout * cin * taps, device='cuda')
            g = lambda: nv.call('iunet_conv3_wgrad', dt, nd, nv.ptr(x), cin * vox, nv.ptr(dy), cout * vox, nv.ptr(slab), nv.ptr(dW), 1.0,
                                a.n, D, S, S, cin, cout, nv.stream())
            ms2 = timeit(g, iters=a.iters)
            line += f' | wgrad {ms2*1e3:8.1f} us {fl/ms2/1e9:7.1f} TF/s'
        if a.f8:
            wb = torch.zeros(nv.lib().iunet_f8_pack_conv3_bytes(cout, cin, taps), dtype=torch.uint8, device='cuda')
            sc = torch.empty(cout, device='cuda')
            nv.call('iunet_f8_pack_conv3', nv.ptr(w), None, None, None, None, 1e-5, nv.ptr(wb), nv.ptr(sc), None, cout, cin, taps, nv.stream())
            need = nv.lib().iunet_conv3_f8_workspace_elems(nd, a.n, D, S, S, cin, cout)
            wk = torch.empty(need, device='cuda') if need else None
            qf = 1 if (a.f8 == 2 and nd == 3 and nv.lib().iunet_f8_pack_order(taps, cin)) else 0
            if qf:       # e4m3 planes: one byte per element (random e4m3 values of moderate size: timing only)
                xq = (torch.randn(a.n * cin * vox, device='cuda') * 2).to(torch.float8_e4m3fn).view(torch.uint8)
                yq = torch.empty(a.n * cout * vox, dtype=torch.uint8, device='cuda')
            h = lambda: nv.call('iunet_conv3_f8_fwd_q', dt, nd, nv.ptr(xq if qf else x), cin * vox, qf, nv.ptr(yq if qf else y), cout * vox, qf,
                                nv.ptr(wb), nv.ptr(sc), nv.ptr(bias), a.n, D, S, S, cin, cout, 2, nv.ptr(wk), nv.stream())
            ms3 = timeit(h, iters=a.iters)
            line += f' | fp8{" (e4m3 planes)" if qf else ""} {ms3*1e3:8.1f} us {fl/ms3/1e9:7.1f} TF/s ({ms/ms3:.2f}x){" split-K" if need else ""}'
        if a.x2:
            wv = torch.empty(3 * cin * cout * taps, device='cuda')
            osc, b2 = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
            nv.call('iunet_x2_prep', nv.ptr(w), nv.ptr(wv), nv.ptr(osc), nv.ptr(b2), None, None, None, None, nv.ptr(bias), 1e-5, 64.0, 64.0,
                    cout, cin, taps, 0, 16 if nd == 3 else 32, nv.stream())
            xpm = nv.lib().iunet_x2_pack_mode(nd)        # padded K16 order in 3-D, compact order (cross-pair step) in 2-D
            wx = torch.empty(nv.pack_conv3_elems(cout, 3 * cin, taps, xpm), dtype=torch.float16, device='cuda')
            nv.call('iunet_pack_conv3', 0, nv.ptr(wv), None, nv.ptr(wx), cout, 3 * cin, taps, xpm, nv.stream())
            xs = (torch.randn(a.n * 2 * cin * vox, device='cuda') * 8).to(torch.float16)      # hi planes | lo planes (random words: timing only)
            ys = torch.empty(a.n * 2 * cout * vox, dtype=torch.float16, device='cuda')
            k = lambda: nv.call('iunet_x2_conv3_fwd', nd, nv.ptr(xs), 2 * cin * vox, cin // 8, nv.ptr(ys), 2 * cout * vox, cout // 8, nv.ptr(wx),
                                nv.ptr(osc), nv.ptr(b2), a.n, D, S, S, cin, cout, 2, nv.stream())
            ms4 = timeit(k, iters=a.iters)
            line += f' | fp16x2 {ms4*1e3:8.1f} us {fl/ms4/1e9:7.1f} TF/s algorithmic = {3*fl/ms4/1e9:7.1f} TF/s of MFMA work ({ms4/ms:.2f}x the 16-bit time)'
        if a.x2m:
            whi = torch.empty(cin * cout * taps, device='cuda')
            w8 = torch.zeros(nv.lib().iunet_x2m_w8_bytes_nd(nd, cout, cin), dtype=torch.uint8, device='cuda')
            osc, b2 = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
            nv.call('iunet_x2m_prep_nd', nd, nv.ptr(w), nv.ptr(whi), nv.ptr(w8), nv.ptr(osc), nv.ptr(b2), None, None, None, None, 1e-5, 64.0, 64.0, cout, cin, nv.stream())
            pm16 = 2 if nd == 3 else 6
            w16 = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pm16), dtype=torch.float16, device='cuda')
            nv.call('iunet_pack_conv3', 0, nv.ptr(whi), None, nv.ptr(w16), cout, cin, taps, pm16, nv.stream())
            xh = (torch.randn(a.n * cin * vox, device='cuda') * 8).to(torch.float16)                  # hi planes
            x8 = (torch.randn(a.n * 2 * cin * vox, device='cuda') * 2).to(torch.float8_e4m3fn).view(torch.uint8)      # lo8 planes (random e4m3 bytes: timing only)
            yh = torch.empty(a.n * cout * vox, dtype=torch.float16, device='cuda')
            y8 = torch.empty(a.n * 2 * cout * vox, dtype=torch.uint8, device='cuda')
            km = lambda: nv.call('iunet_x2m_conv_fwd', nd, nv.ptr(xh), cin * vox, nv.ptr(x8), 2 * cin * vox, nv.ptr(yh), cout * vox, -1, nv.ptr(y8), 2 * cout * vox,
                                 nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b2), a.n, D, S, S, cin, cout, 2, None, nv.stream())
            ms5 = timeit(km, iters=a.iters)
            line += (f' | x2m {ms5*1e3:8.1f} us {fl/ms5/1e9:7.1f} TF/s algorithmic = {2*fl/ms5/1e9:7.1f} TF/s of matrix work in 16-bit equivalents '
                     f'(1 x 16-bit + 2 x fp8 at twice the rate; {ms5/ms:.2f}x the 16-bit time)')
        print(line, flush=True)
    print(f'sum fwd: {tot_t*1e3:.1f} us, {tot_f/tot_t/1e9:.1f} TF/s')

if __name__ == '__main__':
    main()
