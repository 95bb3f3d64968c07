"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the reference's batch
producer (interactive_unet/loader.py).

Follows
  /root/reference/interactive_unet/loader.py:28-44    load_annotations: image/mask/weight -> [C, H, W] float32 in [0, 1] (uint8 / 255
                                                     in float64, then float32), weight repeated over the classes, mask and weight
                                                     zeroed where image[0] == 0
  /root/reference/interactive_unet/loader.py:125-133  the transform chain: RandomHorizontalFlip, RandomVerticalFlip,
                                                     RandomRotation((-360, 360), NEAREST), RandomResizedCrop((512, 512), scale (0.3, 1), NEAREST)
  /root/reference/interactive_unet/loader.py:138-154  __getitem__: the same transform parameters for image, mask and weight; float16 out

The transforms are torchvision.transforms.v2 (torchvision is NOT in /root/reference and NOT in this image: pinned version
unknown -- pyproject.toml names only `torchvision`): their published algorithms are restated here --
  horizontal / vertical flip  = flip of the last / second-to-last axis;
  rotate(angle, NEAREST, expand=False, center=None, fill=None) = affine grid of the inverse rotation about the image centre
      (_get_inverse_affine_matrix with -angle, _affine_grid: base grid linspace((1-W)/2, (W-1)/2, W) x linspace((1-H)/2, (H-1)/2, H),
      theta^T / (W/2, H/2), bmm) + torch.nn.functional.grid_sample(mode='nearest', padding_mode='zeros', align_corners=False);
      angle % 360 in {0, 180} (and {90, 270} for square images) are exact rot90 fast paths;
  resized_crop(top, left, h, w, (512, 512), NEAREST) = crop + torch.nn.functional.interpolate(mode='nearest').
**Parity unpinned at the torchvision boundary.**  What IS pinned: the normalisation (tests/golden/loader.npz, produced by the
reference's own lines 32-42), and the two torch primitives under the transforms -- `transform_reference_ops` below runs the
chain through torch's own grid_sample / interpolate / flip, and `transform` (the explicit per-pixel index arithmetic the HIP
kernel follows, every fp32 operation rounded on its own) is checked against it in tests/test_oracle_golden.py.
"""
import math
import numpy as np
import torch

OUT = 512


def normalise(image_u8, mask_u8, weight_u8):
    """loader.py:32-42 for one annotation.  image [H, W] or [H, W, ch] uint8, mask [H, W, C] uint8 (one-hot x 255), weight
    [H, W] uint8 -> float32 [ch, H, W], [C, H, W], [C, H, W]."""
    image = image_u8[:, :, None] if image_u8.ndim == 2 else image_u8
    weight = np.repeat(weight_u8[:, :, None], mask_u8.shape[-1], axis=2)
    image = (np.moveaxis(image, -1, 0) / 255).astype('float32')
    mask = (np.moveaxis(mask_u8, -1, 0) / 255).astype('float32')
    weight = (np.moveaxis(weight, -1, 0) / 255).astype('float32')
    for c in range(weight.shape[0]):
        weight[c][image[0] == 0] = 0.0
        mask[c][image[0] == 0] = 0.0
    return image, mask, weight


def inverse_rotation_matrix(angle):
    """torchvision _get_inverse_affine_matrix(center=[0, 0], angle=-angle_deg, translate=[0, 0], scale=1, shear=[0, 0]) as rotate calls it."""
    rot = math.radians(-angle)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)     # shear 0: cos(sy) = 1, tan(sx) = 0
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * 0.0 + m[1] * 0.0
    m[5] += m[3] * 0.0 + m[4] * 0.0
    return m


def rescaled_theta(angle, W, H):
    """theta^T / (0.5 W, 0.5 H) in float32, as _affine_grid forms it: [3][2]."""
    theta = torch.tensor(inverse_rotation_matrix(angle), dtype=torch.float32).reshape(1, 2, 3)
    return theta.transpose(1, 2).div(torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32))[0]


def base_grids(W, H):
    return (torch.linspace((1.0 - W) * 0.5, (W - 1.0) * 0.5, steps=W), torch.linspace((1.0 - H) * 0.5, (H - 1.0) * 0.5, steps=H))


def rotation_kind(angle, H, W):
    """0: affine grid; 1: copy; 2: rot90 k=2; 3: rot90 k=1; 4: rot90 k=3 (torchvision rotate fast paths)."""
    a = angle % 360
    if a == 0:
        return 1
    if a == 180:
        return 2
    if H == W and a == 90:
        return 3
    if H == W and a == 270:
        return 4
    return 0


def transform_reference_ops(x, hflip, vflip, angle, crop):
    """The chain on one float32 [C, H, W] tensor through torch's own primitives."""
    import torch.nn.functional as F
    t = torch.as_tensor(x)
    if hflip:
        t = t.flip(-1)
    if vflip:
        t = t.flip(-2)
    C, H, W = t.shape
    kind = rotation_kind(angle, H, W)
    if kind == 1:
        t = t.clone()
    elif kind >= 2:
        t = torch.rot90(t, k={2: 2, 3: 1, 4: 3}[kind], dims=(-2, -1))
    else:
        xg, yg = base_grids(W, H)
        base = torch.empty(1, H, W, 3)
        base[..., 0].copy_(xg)
        base[..., 1].copy_(yg.unsqueeze(-1))
        base[..., 2].fill_(1)
        grid = base.view(1, H * W, 3).bmm(rescaled_theta(angle, W, H).unsqueeze(0)).view(1, H, W, 2)
        t = F.grid_sample(t[None], grid, mode='nearest', padding_mode='zeros', align_corners=False)[0]
    i, j, h, w = crop
    t = t[..., i:i + h, j:j + w]
    return F.interpolate(t[None], size=[OUT, OUT], mode='nearest')[0]


def _fma(a, b, c):
    """float32 fused multiply-add (one rounding): exact in float64 for float32 operands of this magnitude."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def source_index(H, W, hflip, vflip, angle, crop):
    """Explicit per-pixel arithmetic: for every output pixel the source (row, col) in the UNFLIPPED annotation, or -1 where the
    rotation samples outside.  float32, in the order torch's CPU bmm accumulates the K = 3 products (the reference's loader
    runs its transforms on the CPU, loader.py:95-99): g = fma(y, r1, x * r0) + r2 -- bit-identical to the bmm + grid_sample
    path on every pixel tested.  This is the definition the HIP kernel implements."""
    i, j, h, w = crop
    f32 = np.float32
    oy = np.arange(OUT, dtype=np.float32)
    cy = i + np.minimum(np.floor(oy * (f32(h) / f32(OUT))).astype(np.int64), h - 1)          # interpolate(mode='nearest')
    cx = j + np.minimum(np.floor(oy * (f32(w) / f32(OUT))).astype(np.int64), w - 1)
    CY, CX = np.meshgrid(cy, cx, indexing='ij')
    kind = rotation_kind(angle, H, W)
    if kind == 1:
        ry, rx, ok = CY, CX, np.ones_like(CY, bool)
    elif kind == 2:
        ry, rx, ok = H - 1 - CY, W - 1 - CX, np.ones_like(CY, bool)
    elif kind == 3:                      # rot90 k=1: out[y][x] = in[x][W - 1 - y]
        ry, rx, ok = CX, W - 1 - CY, np.ones_like(CY, bool)
    elif kind == 4:                      # rot90 k=3: out[y][x] = in[H - 1 - x][y]
        ry, rx, ok = H - 1 - CX, CY, np.ones_like(CY, bool)
    else:
        xg, yg = [g.numpy() for g in base_grids(W, H)]
        r = rescaled_theta(angle, W, H).numpy()
        X, Y = xg[CX], yg[CY]
        gx = _fma(Y, r[1, 0], X * r[0, 0]) + r[2, 0]
        gy = _fma(Y, r[1, 1], X * r[0, 1]) + r[2, 1]
        ix = ((gx + f32(1)) * f32(W) - f32(1)) / f32(2)
        iy = ((gy + f32(1)) * f32(H) - f32(1)) / f32(2)
        rx, ry = np.rint(ix).astype(np.int64), np.rint(iy).astype(np.int64)
        ok = (rx >= 0) & (rx < W) & (ry >= 0) & (ry < H)
    if vflip:
        ry = H - 1 - ry
    if hflip:
        rx = W - 1 - rx
    return np.where(ok, ry, -1), np.where(ok, rx, -1)


def transform(x, hflip, vflip, angle, crop):
    """The chain as one gather (what the device does)."""
    x = np.asarray(x)
    ry, rx = source_index(x.shape[1], x.shape[2], hflip, vflip, angle, crop)
    out = x[:, np.clip(ry, 0, None), np.clip(rx, 0, None)]
    out[:, ry < 0] = 0
    return out


def resized_crop_params(H, W, rng_uniform, rng_randint, scale=(0.3, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params / make_params: rng_uniform(a, b) -> float, rng_randint(n) -> int in [0, n)."""
    area = H * W
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng_uniform(scale[0], scale[1])
        aspect_ratio = math.exp(rng_uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= W and 0 < h <= H:
            i = rng_randint(H - h + 1)
            j = rng_randint(W - w + 1)
            return i, j, h, w
    in_ratio = float(W) / float(H)
    if in_ratio < min(ratio):
        w = W
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = H
        w = int(round(h * max(ratio)))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


def get_item(image, mask, weight, hflip, vflip, angle, crop, augment=True):
    """loader.py:138-154 with given transform parameters: float32 [C, H, W] annotation -> three float16 [C, 512, 512] (or the
    annotation itself in float16 without augmentation)."""
    outs = []
    for t in (image, mask, weight):
        t = transform(t, hflip, vflip, angle, crop) if augment else np.asarray(t)
        outs.append(torch.from_numpy(np.ascontiguousarray(t)).to(torch.float16))
    return outs
