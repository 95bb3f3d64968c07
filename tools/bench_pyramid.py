"""Time the device multiscale pyramid (interactive_unet.utils.multiscale_levels, iunet_zoom_nearest_u8) on a V^3 uint8
volume and report bytes moved against the HBM roofline; scipy.ndimage.zoom (the call the reference makes per shard,
utils.py:46) is timed on one 256^3 shard for the CPU figure.   python tools/bench_pyramid.py [--size 1024] [--channels 0]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from interactive_unet import utils


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=1024); ap.add_argument('--channels', type=int, default=0)
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    V = (a.size,) * 3 + ((a.channels,) if a.channels else ())
    chunk = (128,) * 3 + ((a.channels,) if a.channels else ()); shard = (256,) * 3 + ((a.channels,) if a.channels else ())
    vol = torch.randint(1, 256, V, dtype=torch.uint8, device='cuda')
    lv = utils.multiscale_levels(vol, chunk, shard)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        lv = utils.multiscale_levels(vol, chunk, shard)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    t0 = time.time()
    for _ in range(a.iters):
        lv = utils.multiscale_levels(vol, chunk, shard)
    torch.cuda.synchronize()
    wall = (time.time() - t0) / a.iters * 1e3
    # algorithmic bytes: every output byte written once + the input cache lines a nearest 0.5x gather touches (every other
    # row and plane, whole 128-byte lines of those rows): in/4 for 3-D volumes; with an innermost channel axis all of each kept row
    out_b = sum(l.numel() for l in lv)
    in_b, n = 0, vol.numel()
    for l in lv:
        in_b += n // 4
        n = l.numel()
    print(f'{V}: {len(lv)} levels {[tuple(l.shape) for l in lv]}: {ms:.3f} ms GPU ({wall:.3f} ms wall), '
          f'{(out_b + in_b) / ms / 1e6:.0f} GB/s algorithmic ({out_b / 1e6:.0f} MB written, {in_b / 1e6:.0f} MB of lines read), '
          f'{vol.numel() / ms / 1e6:.1f} G source voxels/s')
    from scipy import ndimage
    blk = vol[:256, :256, :256].cpu().numpy()
    t0 = time.time(); ndimage.zoom(blk, 0.5, order=0); t2 = time.time() - t0
    print(f'CPU, one {blk.shape} shard through scipy.ndimage.zoom(order=0): {t2 * 1e3:.1f} ms '
          f'({blk.size / t2 / 1e6:.0f} M source voxels/s, 1 thread)')

if __name__ == '__main__':
    main()
