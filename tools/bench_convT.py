"""Transposed conv (k = 2, s = 2) forward alone: time, algorithmic HBM rate and MFMA rate per shape.
python tools/bench_convT.py [iters=30]   (GPU box; IUNET_CONVT_CHUNK4 / IUNET_CONVT_CAP: A/B switches of iunet_convT_launch)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# (label, nd, N, input grid, Cin, Cout, e4m3 planes out)
SHAPES = [('C3 dec0.up', 3, 2, (64, 64, 64), 64, 32, 0), ('C3 dec1.up', 3, 2, (32, 32, 32), 128, 64, 0),
          ('C3 dec2.up', 3, 2, (16, 16, 16), 256, 128, 0),
          ('C5 dec0.up', 3, 1, (64, 64, 64), 128, 64, 1), ('C5 dec0.up 16-bit', 3, 1, (64, 64, 64), 128, 64, 0),
          ('C5 dec1.up', 3, 1, (32, 32, 32), 256, 128, 1), ('C5 dec2.up', 3, 1, (16, 16, 16), 512, 256, 1),
          ('C5 dec3.up', 3, 1, (8, 8, 8), 1024, 512, 1),
          ('C2 dec0.up', 2, 8, (256, 256), 64, 32, 0), ('C2 dec1.up', 2, 8, (128, 128), 128, 64, 0),
          ('C2 dec2.up', 2, 8, (64, 64), 256, 128, 0)]
for label, nd, N, grid, cin, cout, o8 in SHAPES:
    D, H, W = grid if nd == 3 else (1,) + grid
    vin = D * H * W
    vout = vin * 2 ** nd
    x = (torch.randn(N * cin * vin, device='cuda') * 0.5).to(torch.bfloat16)
    w = torch.randn(cin, cout, *([2] * nd), device='cuda') * 0.05
    b = torch.randn(cout, device='cuda') * 0.1
    wpk = torch.empty(w.numel(), dtype=torch.bfloat16, device='cuda')
    nv.call('iunet_pack_convT', 1, nv.ptr(w), nv.ptr(wpk), cin, cout, 2 ** nd, nv.stream())
    y = torch.empty(N * cout * vout * (1 if o8 else 2), dtype=torch.uint8, device='cuda')
    def run():
        if o8:
            nv.call('iunet_convT_fwd_q', 1, nd, nv.ptr(x), cin * vin, nv.ptr(y), cout * vout, nv.ptr(wpk), nv.ptr(b), N, D, H, W, cin, cout, nv.stream())
        else:
            nv.call('iunet_convT_fwd', 1, nd, nv.ptr(x), cin * vin, nv.ptr(y), cout * vout, nv.ptr(wpk), nv.ptr(b), N, D, H, W, cin, cout, nv.stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    byts = N * (cin * vin * 2 + cout * vout * (1 if o8 else 2)) + w.numel() * 2
    flop = 2.0 * N * vin * cin * cout * 2 ** nd
    print(f'{label:20s} {cin:4d}->{cout:4d} @ {N} x {"x".join(map(str, grid)):11s} {"e4m3 out" if o8 else "16-bit  "}: {us:8.1f} us  '
          f'{byts / us / 1e6:7.2f} TB/s  {flop / us / 1e6:7.1f} TF/s', flush=True)
