"""Time the device batch producer (interactive_unet.loader, iunet_augment_batch) on the UI configuration -- batch 8, 512 x 512
annotations, 2 classes -- against the HBM roofline, and the same chain through torch's CPU primitives (what the reference's
num_workers=0 torchvision loader does per sample) for the CPU figure.   python tools/bench_loader.py [--batch 8] [--size 512]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from interactive_unet import loader


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8); ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--classes', type=int, default=2); ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    S, C, B = a.size, a.classes, a.batch
    samples = []
    for _ in range(B):
        img = rng.integers(1, 256, (S, S), dtype=np.uint8)
        mask = (np.eye(C, dtype=np.uint8)[rng.integers(0, C, (S, S))] * 255).astype(np.uint8)
        samples.append((img, mask, rng.integers(0, 256, (S, S), dtype=np.uint8)))
    ds = loader.UNetDataset(loader.annotations_from_arrays(samples), None, augment=True, generator=torch.Generator().manual_seed(0))
    idx = list(range(B))
    params = [loader.draw_params(S, S, ds.generator) for _ in idx]
    for _ in range(3): ds.batch(idx, params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): ds.batch(idx, params)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    t0 = time.time()
    for _ in range(a.iters): ds.batch(idx)                       # with the parameter draw and the descriptor upload
    torch.cuda.synchronize()
    wall = (time.time() - t0) / a.iters * 1e3
    out_b = B * (1 + 2 * C) * 512 * 512 * 2
    in_b = B * 512 * 512 * (1 + C + 1)                           # at most one source pixel per output pixel
    print(f'batch {B} x {S}^2, {C} classes -> 512^2 fp16: {ms * 1e3:.1f} us GPU per batch ({wall * 1e3:.1f} us wall with parameter draw + '
          f'descriptor upload), {(out_b + in_b) / ms / 1e6:.0f} GB/s ({out_b / 1e6:.1f} MB written, <= {in_b / 1e6:.1f} MB gathered), '
          f'{B * 512 * 512 / ms / 1e6:.2f} Gpixel/s')
    # the reference's per-sample CPU work, through torch's own primitives (the transforms of loader.py:125-129 are torchvision's
    # rotate = affine grid + grid_sample(nearest) and resized_crop = crop + interpolate(nearest); torchvision itself is not installed)
    import math
    import torch.nn.functional as F
    img, mask, weight = samples[0]
    f32 = [torch.from_numpy((np.moveaxis(a if a.ndim == 3 else a[:, :, None], -1, 0) / 255).astype('float32')) for a in (img, mask, weight)]

    def chain(t, hflip, vflip, angle, crop):
        if hflip: t = t.flip(-1)
        if vflip: t = t.flip(-2)
        C, H, W = t.shape
        rot = math.radians(-angle)
        theta = torch.tensor([math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]).reshape(1, 2, 3)
        base = torch.empty(1, H, W, 3)
        base[..., 0].copy_(torch.linspace((1.0 - W) * 0.5, (W - 1.0) * 0.5, steps=W))
        base[..., 1].copy_(torch.linspace((1.0 - H) * 0.5, (H - 1.0) * 0.5, steps=H).unsqueeze(-1))
        base[..., 2].fill_(1)
        grid = base.view(1, H * W, 3).bmm(theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H])).view(1, H, W, 2)
        t = F.grid_sample(t[None], grid, mode='nearest', padding_mode='zeros', align_corners=False)[0]
        i, j, h, w = crop
        return F.interpolate(t[None, :, i:i + h, j:j + w], size=[512, 512], mode='nearest')[0].to(torch.float16)

    t0 = time.time()
    n = 5
    for k in range(n):
        for t in f32:
            chain(t, *params[k % B])
    cpu = (time.time() - t0) / n
    print(f'CPU, one sample through torch flip / grid_sample / interpolate ({torch.get_num_threads()} threads): {cpu * 1e3:.1f} ms '
          f'= {512 * 512 / cpu / 1e6:.1f} Mpixel/s; a batch of {B}: {cpu * B * 1e3:.0f} ms')

if __name__ == '__main__':
    main()
