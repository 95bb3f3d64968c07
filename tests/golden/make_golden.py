"""Generate golden vectors from the reference itself (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference/interactive_unet/{metrics,slicer}.py as they are and exec's the
pure helper line ranges of predict.py (79-112 and 270-411; the module itself does not
parse on Python 3.10 and imports zarr) and of utils.py (29-48, resize_volume; the module imports cv2, zarr, numba) and loader.py (32-42, the normalisation
block of load_annotations; the module imports skimage and torchvision).  Writes small .npz fixtures next to this file.
The fixtures hold only inputs and expected outputs -- no reference source text.
The reference never travels to the GPU box; tests there read the .npz files.
"""
import os
import sys
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import interactive_unet.metrics as rm          # noqa: E402
import interactive_unet.slicer as rs           # noqa: E402


def ref_predict_helpers():
    src = open(os.path.join(REF, 'interactive_unet', 'predict.py')).read().split('\n')
    ns = {'np': np, 'torch': torch}
    exec('\n'.join(src[78:112]), ns)      # predict_block            (predict.py:79-112)
    exec('\n'.join(src[269:411]), ns)     # helpers                  (predict.py:270-411)
    return ns


LOSSES = [('ce', rm.crossentropy_loss), ('dice', rm.dice_loss), ('iou', rm.iou_loss),
          ('mcc', rm.mcc_loss), ('dice_ce', rm.dice_ce_loss), ('iou_ce', rm.iou_ce_loss),
          ('mcc_ce', rm.mcc_ce_loss)]


def make_losses():
    rng = np.random.default_rng(0)
    out = {}
    case = 0
    for shape in ((2, 2, 16, 16), (3, 4, 8, 8)):
        for wkind in ('none', 'mask', 'ones', 'soft'):
            for axes in ([2, 3], [0, 2, 3]):
                logits = rng.normal(size=shape).astype(np.float32) * 2
                p = torch.softmax(torch.tensor(logits), dim=1).double()
                lab = rng.integers(0, shape[1], size=(shape[0],) + shape[2:])
                y = np.stack([(lab == c) for c in range(shape[1])], 1).astype(np.float64)
                if wkind == 'none':
                    w = None
                elif wkind == 'mask':
                    w2 = (rng.random((shape[0], 1) + shape[2:]) > 0.4).astype(np.float64)
                    w = np.repeat(w2, shape[1], axis=1)      # loader.py:34: weight repeated over C
                    y = y * w                                # loader.py:40-42 zeroes masked labels
                elif wkind == 'ones':
                    w = np.ones(shape)
                else:
                    w = np.repeat(rng.random((shape[0], 1) + shape[2:]), shape[1], axis=1)
                key = f'c{case}'
                out[key + '_p'] = p.numpy()
                out[key + '_y'] = y
                out[key + '_axes'] = np.array(axes)
                if w is not None:
                    out[key + '_w'] = w
                for name, fn in LOSSES:
                    pt = p.clone().requires_grad_(True)
                    wt = None if w is None else torch.tensor(w)
                    val = fn(pt, torch.tensor(y), wt, axes=axes)
                    val.backward()
                    out[f'{key}_{name}'] = np.float64(val.item())
                    out[f'{key}_{name}_grad'] = pt.grad.numpy()
                # rounded metrics as logged by unet.py:80-86
                yh, yt = torch.round(p), torch.round(torch.tensor(y))
                wt = None if w is None else torch.tensor(w)
                out[key + '_rounded'] = np.array([rm.dice(yh, yt, wt, axes=axes).item(),
                                                  rm.iou(yh, yt, wt, axes=axes).item(),
                                                  rm.mcc(yh, yt, wt, axes=axes).item()])
                case += 1
    out['n_cases'] = np.array(case)
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)
    print('losses.npz:', case, 'cases')


def make_predict():
    h = ref_predict_helpers()
    out = {}
    # G3 block coordinates
    k = 0
    for V in ((1024, 1024, 1024), (300, 260, 129), (128, 128, 128), (97, 200, 513), (72, 72, 72)):
        for S in (128, 256, 130, 32):
            for o in (0.25, 0.5, 0.0):
                if np.prod(np.ceil(np.array(V) / (S * (1 - o) + 1e-9))) > 3000:
                    continue
                b, pb, lb = h['get_block_coordinates'](np.array(V), input_size=S, overlap=o)
                out[f'bc{k}_args'] = np.array(list(V) + [S, int(o * 100)])
                out[f'bc{k}_b'], out[f'bc{k}_pb'], out[f'bc{k}_lb'] = b, pb, lb
                k += 1
    out['n_bc'] = np.array(k)
    # G4 padded blocks
    rng = np.random.default_rng(1)
    vol = rng.integers(0, 256, size=(40, 36, 44), dtype=np.uint8)
    out['pad_vol'] = vol
    coords = [(-8, -8, -8, 16, 16, 16), (24, 20, 30, 48, 44, 54), (-5, 10, 30, 19, 34, 54),
              (8, 8, 8, 32, 32, 32), (30, -6, -3, 54, 18, 21), (-10, 20, 10, 22, 52, 42)]
    out['pad_coords'] = np.array(coords)
    for i, c in enumerate(coords):
        out[f'pad{i}'] = h['get_padded_block'](vol, *c)
    # reflect_index (shadowed first definition's helper, predict.py:270-279)
    idx = np.arange(-30, 50)
    out['reflect_idx'] = idx
    for n in (1, 2, 7, 16):
        out[f'reflect_{n}'] = h['reflect_index'](idx, n)
    # G5 windows
    for S in (8, 16, 32):
        out[f'gauss{S}'] = h['gaussian_3d'](S, sigma=0.125)
        out[f'hann{S}'] = h['hanning_3d'](S)
    g128 = h['gaussian_3d'](128, sigma=0.125)
    out['gauss128_diag'] = np.array([g128[i, i, i] for i in range(128)])
    out['gauss128_line'] = g128[64, 64, :].copy()
    out['gauss128_minmax'] = np.array([g128.min(), g128.max()])
    # G8 shard coordinates
    out['shards_300_260_129_128'] = h['get_shard_coordinates'](np.array([300, 260, 129]), shard_size=128)
    out['shards_512_256'] = h['get_shard_coordinates'](np.array([512, 512, 512]), shard_size=256)

    # G6 predict_block with deterministic stub models
    class Stub:
        device = torch.device('cpu')

        def __init__(self, kind):
            self.kind = kind

        def __call__(self, x):                       # x [B,1,S,S]
            B, _, H, W = x.shape
            if self.kind == 0:
                l = torch.cat([x, 1 - x], 1)
            else:                                    # asymmetric in rows / cols: catches transposes
                r = torch.arange(H, dtype=x.dtype).view(1, 1, H, 1) / H
                c = torch.arange(W, dtype=x.dtype).view(1, 1, 1, W) / W
                l = torch.cat([x * (1 + r), x * (0.5 + 2 * c) - 0.3 * r, 0.2 + 0 * x], 1)
            return torch.softmax(l, dim=1)
    k = 0
    for S, bs, axes, kind in ((16, 4, [0, 1, 2], 0), (16, 5, [0, 1, 2], 1), (16, 8, [0], 1),
                              (32, 8, [0, 1, 2], 1), (16, 16, [2, 1], 1)):
        blk = rng.random((S, S, S)).astype(np.float32)
        ncls = 2 if kind == 0 else 3
        res = h['predict_block'](Stub(kind), torch.tensor(blk), num_classes=ncls, batch_size=bs, axes=axes)
        out[f'pb{k}_args'] = np.array([S, bs, kind, ncls])
        out[f'pb{k}_axes'] = np.array(axes)
        out[f'pb{k}_block'] = blk
        out[f'pb{k}_out'] = res
        k += 1
    out['n_pb'] = np.array(k)

    # G7 blend + normalise + quantise on an in-memory volume: predict.py:201-256 restated
    # with the reference's own helpers and the arithmetic of lines 244-245 and 255.
    V = (72, 56, 40)
    S = 32
    volume = rng.integers(0, 256, size=V, dtype=np.uint8)
    stub = Stub(1)
    window = h['gaussian_3d'](S, sigma=0.125)
    b, pb, lb = h['get_block_coordinates'](np.array(V), input_size=S, overlap=0.25)
    pred = np.zeros(V + (3,), np.float32)
    weight = np.zeros(V, np.float32)
    for i in range(len(pb)):
        padded = torch.tensor(h['get_padded_block'](volume, *pb[i]).astype('float32') / 255.0)
        P = h['predict_block'](stub, padded, num_classes=3, batch_size=8, axes=[0, 1, 2])
        i0, j0, k0, i1, j1, k1 = b[i]
        a0, b0, c0, a1, b1, c1 = lb[i]
        pred[i0:i1, j0:j1, k0:k1] += P[a0:a1, b0:b1, c0:c1, :] * window[a0:a1, b0:b1, c0:c1, None]
        weight[i0:i1, j0:j1, k0:k1] += window[a0:a1, b0:b1, c0:c1]
    final = (255 * pred / np.maximum(weight, 1e-3)[..., None]).astype('uint8')
    out['blend_volume'] = volume
    out['blend_final'] = final
    out['blend_weight'] = weight
    out['blend_pred_sample'] = pred[::7, ::5, ::3].copy()
    np.savez_compressed(os.path.join(HERE, 'predict.npz'), **out)
    print('predict.npz done')


def make_slicer():
    out = {}
    rng = np.random.default_rng(3)
    vecs = [np.array([1, 0, 0]), np.array([0, 1, 0]), np.array([0, 0, 1]),
            np.array([0.3, -0.5, 0.8]), np.array([-0.7, 0.1, 0.2])]
    vol = (np.arange(32 ** 3) % 251).astype(np.uint8).reshape(32, 32, 32)
    ramp = (rng.random((32, 32, 32)) * 255).astype(np.uint8)
    out['vol'], out['ramp'] = vol, ramp
    for i, rv in enumerate(vecs):
        s = rs.Slicer(volume_shape=[32, 32, 32])
        s.update_orientation_vectors(rv)
        s.origin = np.array([15.5, 14.0, 17.25])
        out[f's{i}_rv'] = rv.astype(float)
        out[f's{i}_origin'] = s.origin.copy()
        out[f's{i}_rotvec'], out[f's{i}_rotmat'] = s.rot_vec, s.rot_mat
        out[f's{i}_u'], out[f's{i}_v'], out[f's{i}_w'] = s.u, s.v, s.w
        out[f's{i}_coords8'] = s.get_interpolation_coords(8)
        for axis in (0, 1, 2):
            for order in (0, 1):
                out[f's{i}_slice_a{axis}_o{order}'] = s.get_slice(ramp, axis=axis, slice_width=24, order=order)
        data = (rng.random((16, 16)) * 255).astype(np.uint8)
        out[f's{i}_upd_data'] = data
        out[f's{i}_upd_vol'] = s.update_volume(data, vol.copy(), axis=1)
        d = s.to_dict()
        s2 = rs.Slicer(volume_shape=[8, 8, 8])
        s2.from_dict(d)
        out[f's{i}_rt_u'] = s2.u
    out['n'] = np.array(len(vecs))
    np.savez_compressed(os.path.join(HERE, 'slicer.npz'), **out)
    print('slicer.npz done')


def ref_resize_volume():
    from scipy import ndimage
    src = open(os.path.join(REF, 'interactive_unet', 'utils.py')).read().split('\n')
    ns = {'np': np, 'ndimage': ndimage}
    exec('\n'.join(src[28:48]), ns)       # resize_volume           (utils.py:29-48)
    return ns['resize_volume']


def make_multiscale():
    """resize_volume of the reference on small uint8 volumes: several blocks per axis, ragged last blocks, sizes whose
    last sample scipy fills with 0 (32, 48, 56), a 4-D [V, V, V, C] prediction volume (the channel axis is zoomed too)."""
    rv = ref_resize_volume()
    rng = np.random.default_rng(5)
    out = {}
    cases = [((40, 32, 70), 16), ((64, 48, 56), 32), ((50, 50, 50), 512), ((24, 24, 24, 2), 8), ((32, 20, 36, 4), 16)]
    for i, (shape, block) in enumerate(cases):
        src = rng.integers(1, 256, shape, dtype=np.uint8)          # never 0: scipy's constant fill is visible
        dst = np.full(tuple(int(x * 0.5) for x in shape), 7, dtype=np.uint8)
        rv(src, dst, scale=0.5, block_size=block, order=0)
        out[f'c{i}_src'], out[f'c{i}_dst'], out[f'c{i}_block'] = src, dst, np.array(block)
    out['n'] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, 'multiscale.npz'), **out)
    print('multiscale.npz done')


def make_loader():
    """The normalisation block of load_annotations (loader.py:32-42, the body of its per-file loop) exec'd on small uint8
    annotations: 2-D and multi-channel images, zeros in the image (mask / weight zeroing), 2 and 3 classes."""
    import textwrap
    src = open(os.path.join(REF, 'interactive_unet', 'loader.py')).read().split('\n')
    first = next(k for k, l in enumerate(src) if 'if len(image_slice.shape) == 2:' in l)
    last = next(k for k, l in enumerate(src) if 'mask_slice[c][image_slice[0] == 0] = 0.0' in l)
    assert (first, last) == (31, 41), (first, last)                      # loader.py:32-42
    block = textwrap.dedent('\n'.join(src[first:last + 1]))
    rng = np.random.default_rng(9)
    out = {}
    cases = [((24, 20), 2, None), ((17, 31), 3, None), ((16, 16), 2, 3)]
    for k, (hw, C, ch) in enumerate(cases):
        image = rng.integers(0, 256, hw if ch is None else hw + (ch,), dtype=np.uint8)
        image[rng.random(image.shape) < 0.2] = 0
        cls = rng.integers(0, C, hw)
        mask = (np.eye(C, dtype=np.uint8)[cls] * 255).astype(np.uint8)
        weight = rng.integers(0, 256, hw, dtype=np.uint8)
        ns = {'np': np, 'image_slice': image.copy(), 'mask_slice': mask.copy(), 'weight_slice': weight.copy()}
        exec(block, ns)
        out[f'c{k}_image'], out[f'c{k}_mask'], out[f'c{k}_weight'] = image, mask, weight
        out[f'c{k}_image_f'], out[f'c{k}_mask_f'], out[f'c{k}_weight_f'] = ns['image_slice'], ns['mask_slice'], ns['weight_slice']
    out['n'] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, 'loader.npz'), **out)
    print('loader.npz done')


if __name__ == '__main__':
    make_loader()
    make_multiscale()
    make_losses()
    make_predict()
    make_slicer()
