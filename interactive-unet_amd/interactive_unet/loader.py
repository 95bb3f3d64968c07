"""Drop-in for interactive_unet/loader.py ("Batch producer"): `load_annotations`,
`get_data_loader`, `UNetDataset` with the reference's names and arguments.  The annotations stay uint8 and live on the GPU;
every batch -- normalisation (loader.py:32-42), RandomHorizontalFlip / RandomVerticalFlip / RandomRotation(NEAREST) /
RandomResizedCrop((512, 512), NEAREST) (loader.py:125-133) and the float16 conversion (loader.py:150-152) -- is ONE gather
launch (libiunet: iunet_augment_batch).  The random parameters are drawn on the host from torch's generator in the order
torchvision's v2 transforms draw them (flip, flip, angle, crop box); the reference runs the same chain per sample on the CPU
with num_workers=0 (loader.py:95-99).

File reading (TIFF / PNG through PIL) and `colored_to_categorical` are caller-side format code; `annotations_from_arrays`
takes arrays directly.  `reslice=True` (load_resliced_annotations, dead code in the reference: trainer.py:18) is not provided.
"""
import ctypes
import glob
import math
import os

import numpy as np
import torch

from . import _native as nv

OUT_SIZE = 512                      # RandomResizedCrop(size=(512, 512)) ignores input_size (loader.py:128)
COLORS = np.array([[0, 0, 0], [230, 25, 75], [60, 180, 75], [255, 225, 25], [0, 130, 200], [245, 130, 48], [145, 30, 180],
                   [70, 240, 240], [240, 50, 230], [210, 245, 60], [170, 255, 195]], dtype=np.uint8)      # utils.py:304-306


class AugDesc(ctypes.Structure):
    """Mirror of csrc/augment.hip: AugDesc."""
    _fields_ = [('image', ctypes.c_void_p), ('mask', ctypes.c_void_p), ('weight', ctypes.c_void_p), ('xg', ctypes.c_void_p),
                ('yg', ctypes.c_void_p), ('H', ctypes.c_int), ('W', ctypes.c_int), ('hflip', ctypes.c_int), ('vflip', ctypes.c_int),
                ('kind', ctypes.c_int), ('ci', ctypes.c_int), ('cj', ctypes.c_int), ('ch', ctypes.c_int), ('cw', ctypes.c_int),
                ('r', ctypes.c_float * 6), ('keep_dark', ctypes.c_int)]


def colored_to_categorical(colored_mask):
    """utils.py:308-349: one-hot x 255 over the palette colours present in the mask (first match), background channel split
    off as weight = 255 - background."""
    flat = colored_mask.reshape(-1, 3).astype(np.uint32)
    keys = flat[:, 0] << 16 | flat[:, 1] << 8 | flat[:, 2]
    ckeys = COLORS[:, 0].astype(np.uint32) << 16 | COLORS[:, 1].astype(np.uint32) << 8 | COLORS[:, 2]
    present = ckeys[np.isin(ckeys, keys)]
    mask = np.zeros(colored_mask.shape[:2] + (len(present),), dtype=np.uint8)
    todo = np.ones(len(keys), bool)
    for k, key in enumerate(present):
        hit = todo & (keys == key)
        mask.reshape(-1, len(present))[hit, k] = 255
        todo &= ~hit
    return mask[:, :, 1:], 255 - mask[:, :, 0]


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def annotations_from_arrays(samples, device='cuda'):
    """samples: iterable of (image uint8 [H, W] or [H, W, ch], mask uint8 [H, W, C] one-hot x 255, weight uint8 [H, W]).
    Returns the annotation list the dataset works on: uint8 tensors resident on the GPU (normalisation happens in the batch
    kernel, so nothing is expanded to float32 here as loader.py:37-39 does)."""
    out = []
    for image, mask, weight in samples:
        image = np.asarray(image)
        image = image[:, :, None] if image.ndim == 2 else image
        t = [torch.from_numpy(np.array(a, dtype=np.uint8, order='C')).to(device) for a in (image, mask, weight)]
        if t[1].shape[:2] != t[0].shape[:2] or t[2].shape != t[0].shape[:2]:
            raise ValueError(f'annotation shapes differ: image {tuple(t[0].shape)}, mask {tuple(t[1].shape)}, weight {tuple(t[2].shape)}')
        out.append(t)
    return out


def load_annotations(set_type='train', device='cuda'):
    """loader.py:15-46: data/{train,val}/{images,masks,weights}/* in sorted order."""
    folder = os.path.join('data', 'train' if set_type == 'train' else 'val')
    names = [np.sort(glob.glob(os.path.join(folder, sub, '*'))) for sub in ('images', 'masks', 'weights')]
    samples = []
    for fi, fm, fw in zip(*names):
        mask, _ = colored_to_categorical(_imread(fm))
        samples.append((_imread(fi), mask, _imread(fw)))
    return annotations_from_arrays(samples, device)


# ---- random parameters, in the order torchvision's v2 transforms draw them ------------------------------------------------
def _uniform(a, b, gen):
    return torch.empty(1).uniform_(a, b, generator=gen).item()


def resized_crop_params(H, W, gen, scale=(0.3, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.make_params: (top, left, height, width)."""
    area = H * W
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * _uniform(scale[0], scale[1], gen)
        aspect_ratio = math.exp(_uniform(log_ratio[0], log_ratio[1], gen))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= W and 0 < h <= H:
            i = torch.randint(0, H - h + 1, size=(1,), generator=gen).item()
            j = torch.randint(0, W - w + 1, size=(1,), generator=gen).item()
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


def draw_params(H, W, gen=None):
    """One sample's (hflip, vflip, angle, crop) for the chain of loader.py:125-129."""
    hflip = bool(torch.rand(1, generator=gen).item() < 0.5)
    vflip = bool(torch.rand(1, generator=gen).item() < 0.5)
    angle = _uniform(-360.0, 360.0, gen)
    return hflip, vflip, angle, resized_crop_params(H, W, gen)


def _rotation(angle, H, W):
    """(kind, r[6]) of torchvision's rotate(angle, NEAREST, expand=False, center=None): the fast paths for multiples of 90
    degrees, else theta^T / (W/2, H/2) of the inverse rotation about the centre, formed in float32 as _affine_grid does."""
    a = angle % 360
    if a == 0:
        return 1, [0.0] * 6
    if a == 180:
        return 2, [0.0] * 6
    if H == W and a in (90, 270):
        return (3 if a == 90 else 4), [0.0] * 6
    rot = math.radians(-angle)
    m = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]           # [d, -b, 0, -c, a, 0] of the inverse matrix
    theta = torch.tensor(m, dtype=torch.float32).reshape(1, 2, 3)
    r = theta.transpose(1, 2).div(torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32))[0]     # [3][2]
    return 0, [float(r[0, 0]), float(r[1, 0]), float(r[2, 0]), float(r[0, 1]), float(r[1, 1]), float(r[2, 1])]


class UNetDataset:
    """loader.py:103-154.  `annotations`: the list from load_annotations / annotations_from_arrays."""

    def __init__(self, annotations, resliced_annotations=None, reslice=False, reslice_factor=2, augment=False, generator=None,
                 out_size=OUT_SIZE, keep_dark=False):
        if reslice:
            raise NotImplementedError('reslice=True (load_resliced_annotations) is not provided; the reference never enables it')
        self.annotations = annotations
        self.resliced_annotations = resliced_annotations
        self.reslice, self.reslice_factor, self.augment = reslice, reslice_factor, augment
        self.generator = generator
        # out_size: the augmented output (the reference's loader: 512 x 512 always); keep_dark: do not zero mask / weight where
        # the image is 0 (the Suggestor's tensors, suggestor.py:60-65, carry no such masking)
        self.out_size = (out_size, out_size) if isinstance(out_size, int) else tuple(out_size)
        self.keep_dark = bool(keep_dark)
        self._grids, self._lut = {}, None

    def __len__(self):
        return len(self.annotations)

    def _grid(self, n, device):
        key = (n, str(device))
        if key not in self._grids:
            self._grids[key] = torch.linspace((1.0 - n) * 0.5, (n - 1.0) * 0.5, steps=n).to(device)
        return self._grids[key]

    def batch(self, indices, params=None):
        """The batch loader.py's DataLoader would collate from __getitem__(i) for i in indices, in one launch.  `params`:
        optional list of (hflip, vflip, angle, crop) per sample (drawn from the generator when None)."""
        ann = [self.annotations[i] for i in indices]
        dev = ann[0][0].device
        ch, C = int(ann[0][0].shape[2]), int(ann[0][1].shape[2])
        if self.augment:
            OH, OW = self.out_size
        else:
            OH, OW = int(ann[0][0].shape[0]), int(ann[0][0].shape[1])
        descs = (AugDesc * len(ann))()
        for k, (image, mask, weight) in enumerate(ann):
            H, W = int(image.shape[0]), int(image.shape[1])
            if int(image.shape[2]) != ch or int(mask.shape[2]) != C or (not self.augment and (H, W) != (OH, OW)):
                raise RuntimeError('stack expects each tensor to be equal size')      # what default_collate raises in the reference
            if self.augment:
                hflip, vflip, angle, crop = params[k] if params is not None else draw_params(H, W, self.generator)
                kind, r = _rotation(angle, H, W)
            else:
                hflip, vflip, kind, r, crop = False, False, 1, [0.0] * 6, (0, 0, H, W)
            d = descs[k]
            d.image, d.mask, d.weight = image.data_ptr(), mask.data_ptr(), weight.data_ptr()
            d.xg, d.yg = self._grid(W, dev).data_ptr(), self._grid(H, dev).data_ptr()
            d.H, d.W, d.hflip, d.vflip, d.kind = H, W, int(hflip), int(vflip), kind
            d.ci, d.cj, d.ch, d.cw = [int(v) for v in crop]
            d.keep_dark = int(self.keep_dark)
            for q in range(6):
                d.r[q] = r[q]
        if self._lut is None or self._lut.device != dev:
            self._lut = torch.from_numpy((np.arange(256) / 255).astype('float32')).to(torch.float16).to(dev)   # loader.py:37-39, :150
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        B = len(ann)
        X = torch.empty((B, ch, OH, OW), dtype=torch.float16, device=dev)
        y = torch.empty((B, C, OH, OW), dtype=torch.float16, device=dev)
        w = torch.empty((B, C, OH, OW), dtype=torch.float16, device=dev)
        with torch.cuda.device(dev):
            nv.call('iunet_augment_batch', nv.ptr(raw), B, ch, C, OH, OW, nv.ptr(self._lut), nv.ptr(X), nv.ptr(y), nv.ptr(w), nv.stream())
        return X, y, w

    def __getitem__(self, idx):
        X, y, w = self.batch([idx])
        return X[0], y[0], w[0]


class DeviceLoader:
    """What DataLoader(dataset, batch_size, shuffle, num_workers=0) yields (loader.py:95-99), produced on the device."""

    def __init__(self, dataset, batch_size=1, shuffle=False, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, int(batch_size), shuffle, generator

    def __len__(self):
        return -(-len(self.dataset) // self.batch_size)

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for s in range(0, n, self.batch_size):
            yield self.dataset.batch(order[s:s + self.batch_size])


def get_data_loader(set_type='train', num_classes=2, batch_size=2, reslice=False, reslice_factor=2, augment=True, shuffle=True,
                    annotations=None, generator=None):
    """loader.py:84-101 (`annotations`: skip the file read and use these)."""
    if annotations is None:
        annotations = load_annotations(set_type=set_type)
    dataset = UNetDataset(annotations, None, reslice=reslice, reslice_factor=reslice_factor, augment=augment, generator=generator)
    return DeviceLoader(dataset, batch_size=batch_size, shuffle=shuffle, generator=generator)
