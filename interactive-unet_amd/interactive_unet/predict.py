"""Drop-in for interactive_unet/predict.py: same entry points and helper names, device path
native (libiunet).  The float32 `pred` / `weight` accumulators that the reference keeps in
temporary Zarr arrays on disk (predict.py:183-199, :244-245) live in HBM; blocks are cut from
a device-resident uint8 volume with reflect padding on the device (predict.py:291-316).

Host-side integer logic (block grid, shard grid, windows) follows the reference's
arithmetic exactly, including the float -> int truncation of predict.py:390-391.
"""
import glob
import os
import time

import numpy as np
import torch

from . import _native as nv
from . import multiscale


# --------------------------------------------------------------------------- helpers (predict.py:270-411)
def reflect_index(idx, size):
    """predict.py:270-279."""
    idx = np.asarray(idx)
    if size == 1:
        return np.zeros_like(idx)
    period = 2 * size - 2
    idx = np.abs(idx) % period
    return np.where(idx < size, idx, period - idx)


def get_padded_block(volume, i0, j0, k0, i1, j1, k1):
    """predict.py:291-316 for host arrays (numpy / zarr-like): clip, read, reflect-pad."""
    vs = volume.shape
    before = [max(0, -i0), max(0, -j0), max(0, -k0)]
    after = [max(0, i1 - vs[0]), max(0, j1 - vs[1]), max(0, k1 - vs[2])]
    block = np.asarray(volume[max(i0, 0):min(i1, vs[0]), max(j0, 0):min(j1, vs[1]), max(k0, 0):min(k1, vs[2])])
    return np.pad(block, tuple(zip(before, after)), mode='reflect')


def get_shard_coordinates(volume_shape, shard_size=128):
    """predict.py:318-325."""
    volume_shape = np.asarray(volume_shape)
    starts = [np.arange(0, s, shard_size) for s in volume_shape]
    c = np.stack(np.meshgrid(*starts, indexing='ij'), -1).reshape(-1, 3)
    return np.concatenate([c, np.minimum(c + shard_size, volume_shape)], axis=1)


def gaussian_3d(input_size, sigma=0.125, eps=1e-3):
    """predict.py:327-347 (float32 arithmetic throughout)."""
    sigma *= input_size
    coords = np.arange(input_size, dtype=np.float32) - (input_size - 1) / 2.0
    g = np.exp(-(coords ** 2) / (2 * sigma ** 2)).astype(np.float32)
    g /= g.max()
    gaussian = g[:, None, None] * g[None, :, None] * g[None, None, :]
    gaussian /= gaussian.max()
    return np.clip(gaussian, max(gaussian.min(), eps), 1.0)


def hanning_3d(input_size, eps=1e-3):
    """predict.py:349-360."""
    h = np.hanning(input_size)
    hanning = h[:, None, None] * h[None, :, None] * h[None, None, :]
    hanning /= hanning.max()
    return np.clip(hanning, max(hanning.min(), eps), 1.0).astype('float32')


def get_block_coordinates(volume_shape, input_size=256, overlap=0.25):
    """predict.py:362-411: (clipped block, padded block, local) coordinates, int arrays [n, 6]."""
    volume_shape = np.asarray(volume_shape)
    n = np.ceil((volume_shape - overlap * input_size) / (input_size - overlap * input_size)).astype(int)
    padded_shape = np.round(n * input_size - (n - 1) * input_size * overlap).astype(int)
    shift = (padded_shape - volume_shape) // 2
    shift6 = np.concatenate([shift, shift])
    step = input_size * (1 - overlap)
    blocks, pblocks, locs = [], [], []
    for i in range(n[0]):
        for j in range(n[1]):
            for k in range(n[2]):
                lo = [i * input_size * (1 - overlap), j * input_size * (1 - overlap), k * input_size * (1 - overlap)]
                c = (np.array(lo + [v + input_size for v in lo]) - shift6).astype(int)     # truncation
                pblocks.append(c)
                lo_c = np.maximum(c[:3], 0)
                hi_c = np.minimum(c[3:], volume_shape)
                blocks.append(np.concatenate([lo_c, hi_c]))
                locs.append(np.concatenate([lo_c - c[:3], hi_c - c[:3]]))
    del step
    return np.array(blocks), np.array(pblocks), np.array(locs)


# --------------------------------------------------------------------------- device accumulators
class VolumeAccumulator:
    """pred [V, C] / weight [V] float32 and the final uint8 [V, C] in HBM (replaces the temp
    Zarr arrays of predict.py:183-199) + the Gaussian window and one block of probabilities."""

    def __init__(self, volume_shape, num_classes, input_size, device, window=None):
        self.V = tuple(int(v) for v in volume_shape)
        self.C, self.S = int(num_classes), int(input_size)
        self.device = torch.device(device)
        self.pred = torch.zeros(self.V + (self.C,), dtype=torch.float32, device=self.device)
        self.weight = torch.zeros(self.V, dtype=torch.float32, device=self.device)
        self.final = torch.empty(self.V + (self.C,), dtype=torch.uint8, device=self.device)
        w = gaussian_3d(self.S, sigma=0.125) if window is None else window      # predict.py:153
        self.window = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(self.device)
        self.block_probs = torch.empty((self.S,) * 3 + (self.C,), dtype=torch.float32, device=self.device)
        self._batch = None

    def batch_buffers(self, nb):
        """uint8 [nb, S, S, S] blocks and fp32 [nb, S, S, S, C] probabilities for a batched 3-D forward."""
        if self._batch is None or self._batch[0].shape[0] < nb:
            self._batch = (torch.empty((nb,) + (self.S,) * 3, dtype=torch.uint8, device=self.device),
                           torch.empty((nb,) + (self.S,) * 3 + (self.C,), dtype=torch.float32, device=self.device))
        return self._batch

    def reset(self):
        self.pred.zero_()
        self.weight.zero_()

    def blend(self, block, local, probs=None):
        """predict.py:244-245 for one block (clipped volume coords, coords inside the block)."""
        P = self.block_probs if probs is None else probs
        nv.call('iunet_blend_accumulate', nv.ptr(self.pred), nv.ptr(self.weight), nv.ptr(P), nv.ptr(self.window),
                self.V[0], self.V[1], self.V[2], self.C, self.S, nv.int_array(block), nv.int_array(local), nv.stream())

    def finalize(self, eps=1e-3):
        """predict.py:252-256: uint8(255 * pred / max(weight, eps)), truncating."""
        if self.V[0] * self.V[1] * self.V[2] == 0:
            return self.final
        nv.call('iunet_normalize_quantize', nv.ptr(self.pred), nv.ptr(self.weight), nv.ptr(self.final),
                self.V[0] * self.V[1] * self.V[2], self.C, float(eps), nv.stream())
        return self.final


def gather_block(volume_dev, padded_coords, S, out=None):
    """get_padded_block on a device-resident uint8 volume -> uint8 [S, S, S] on the device."""
    V = volume_dev.shape
    if out is None:
        out = torch.empty((S, S, S), dtype=torch.uint8, device=volume_dev.device)
    nv.call('iunet_gather_block', nv.ptr(volume_dev), V[0], V[1], V[2], int(padded_coords[0]), int(padded_coords[1]),
            int(padded_coords[2]), S, nv.ptr(out), nv.stream())
    return out


# --------------------------------------------------------------------------- block prediction
BLOCK_BATCH = 2      # 3-D blocks per forward: the deep levels (16^3, 32^3 per block) fill the chip better, half the launches (4: no further gain)


def predict_blocks_3d(eng, acc, volume, blocks, padded, local, lo, hi, batch=None):
    """Blocks lo..hi-1 of the flat list through the 3-D net, `batch` at a time (gather -> forward + softmax -> blend);
    the blends happen in list order, so the accumulators are those of the one-block-at-a-time loop."""
    S, C = acc.S, acc.C
    B = max(1, int(batch or BLOCK_BATCH))
    blk, probs = acc.batch_buffers(B)
    for i in range(lo, hi, B):
        nb = min(B, hi - i)
        for j in range(nb):
            gather_block(volume, padded[i + j], S, out=blk[j])
        eng.infer(blk, (S ** 3, S ** 3, S * S, S, 1), nb, S, S, S, probs=probs,
                  out_strides=(S ** 3 * C, 1, S * S * C, S * C, C))
        for j in range(nb):
            acc.blend(blocks[i + j], local[i + j], probs=probs[j])


def predict_block_device(model, block, out, num_classes=2, batch_size=None, axes=(0, 1, 2)):
    """2.5-D prediction of one S^3 block entirely on the device (predict.py:79-112): for each
    axis the 2-D net runs over the slices along that axis -- strided views of the same block,
    no transposed copies -- and the head adds the probabilities into `out` [S,S,S,C] in the
    block's own orientation; the last axis divides by len(axes) (predict.py:110)."""
    eng = model.engine('eval')
    S, C = block.shape[0], num_classes
    if eng.dim != 2:
        raise ValueError('2.5-D block prediction needs a 2-D network')
    bs = S if not batch_size else min(int(batch_size), S)
    sb = (S * S, S, 1)                        # block strides z, y, x
    so = (S * S * C, S * C, C)
    rows_cols = {0: (1, 2), 1: (0, 2), 2: (0, 1)}
    axes = list(axes)
    if bs >= S and len(axes) > 1 and hasattr(eng, 'infer_views') and not os.environ.get('IUNET_2P5D_SEQUENTIAL'):
        # every axis' S slices as ONE batch of len(axes) * S (engine_x2.EngineX2.infer_views): each axis is a strided view of the block
        # for the first conv and of `out` for the head; the network between them runs once.  The outputs are written in axis order
        # (the first axis writes, the others accumulate, the last divides: predict.py:101-110) -- the same bits as the loop below.
        views, outs = [], []
        for ai, axis in enumerate(axes):
            r, c = rows_cols[axis]
            views.append((block.reshape(-1), (sb[axis], 0, 0, sb[r], sb[c]), S))
            outs.append(dict(probs=out.reshape(-1), out_strides=(so[axis], 1, 0, so[r], so[c]), accumulate=(ai > 0),
                             divisor=float(len(axes)) if ai == len(axes) - 1 else 1.0))
        eng.infer_views(views, 1, S, S, outs)
        return out
    for ai, axis in enumerate(axes):
        r, c = rows_cols[axis]
        last = ai == len(axes) - 1
        for i in range(0, S, bs):
            n = min(bs, S - i)
            xin = block.reshape(-1)[i * sb[axis]:]
            oview = out.reshape(-1)[i * so[axis]:]
            eng.infer(xin, (sb[axis], 0, 0, sb[r], sb[c]), n, 1, S, S, probs=oview,
                      out_strides=(so[axis], 1, 0, so[r], so[c]), accumulate=(ai > 0),
                      divisor=float(len(axes)) if last else 1.0)
    return out


def predict_block(model, block, num_classes=2, batch_size=8, axes=[0, 1, 2]):
    """Reference signature (predict.py:79): `block` float tensor [S,S,S] in [0,1] ->
    numpy float32 [S,S,S,C]."""
    dev = model.device
    blk = block.to(dev, torch.float32).contiguous()
    S = blk.shape[0]
    out = torch.empty((S, S, S, num_classes), dtype=torch.float32, device=dev)
    eng = model.engine('eval')
    run = lambda: predict_block_device(model, blk, out, num_classes, batch_size, axes)
    if hasattr(eng, 'run_checked'):
        eng.run_checked(run)
    else:
        run()
    return out.cpu().numpy()


def find_max_batch_size(model, input_size=256, start=4, max_limit=512):
    """predict.py:49-77 probes by doubling until OOM; the native engine sizes its workspace
    analytically, so answer from free HBM instead (same return contract: a power-of-two batch)."""
    eng = model.engine('eval')
    free, _ = torch.cuda.mem_get_info(model.device)
    per_slice = 2 * input_size * input_size * sum(5 * c // (4 ** l) for l, c in enumerate(eng.ch)) * 2
    best = start
    while best * 2 <= max_limit and best * 2 * per_slice < 0.5 * free:
        best *= 2
    return best


# --------------------------------------------------------------------------- entry points
COLORS = np.array([[0, 0, 0], [230, 25, 75], [60, 180, 75], [255, 225, 25], [0, 130, 200], [245, 130, 48],
                   [145, 30, 180], [70, 240, 240], [240, 50, 230], [210, 245, 60], [170, 255, 195]], dtype=np.uint8)


def _load_model(num_channels, num_classes, device):
    from . import unet
    model_path = os.path.join('model', 'model.ckpt')
    if os.path.isfile(model_path):
        model = unet.UNet.load_from_checkpoint(checkpoint_path=model_path).to(device)     # predict.py:22-24
    else:
        model = unet.UNet(num_channels=num_channels, num_classes=num_classes).to(device)
    model.eval()
    return model


_PALETTES = {}


def _palette(device, ncls):
    key = (str(device), ncls)
    if key not in _PALETTES:
        _PALETTES[key] = torch.tensor(np.asarray(COLORS[1:ncls + 1], dtype=np.uint8)).to(device).contiguous()
    return _PALETTES[key]


def predict_slice(image_slice, num_channels=1, num_classes=2, return_probabilities=False, model=None):
    """predict.py:16-47: uint8 [H,W] -> palette-coloured uint8 [H,W,3] (or probabilities
    [1,H,W,C]).  /255, forward, argmax over the first num_classes channels, one-hot*255,
    colours of utils.py:304-306."""
    device = torch.device('cuda')
    if model is None:
        model = _load_model(num_channels, num_classes, device)
    eng = model.engine('eval')
    H, W = image_slice.shape[:2]
    x = torch.as_tensor(np.ascontiguousarray(image_slice)).to(device)
    # the probabilities are written only for the callers that read them (the coloured class map needs the argmax alone)
    need_probs = return_probabilities or num_classes < eng.ncls
    probs = torch.empty((1, eng.ncls, H, W), dtype=torch.float32, device=device) if need_probs else None
    cls = torch.empty((1, H * W), dtype=torch.uint8, device=device)
    if x.dtype != torch.uint8:
        x = (x.to(torch.float32) / 255).contiguous()
    run = lambda: eng.infer(x, (H * W, H * W, H * W, W, 1), 1, 1, H, W, probs=probs, cls=cls)
    if hasattr(eng, 'run_checked'):
        eng.run_checked(run)            # default mode: a forward whose activations left the fp16 range is re-run in a wider form
    else:
        run()
    if return_probabilities:
        return np.moveaxis(probs.cpu().numpy(), 1, -1)
    if num_classes < eng.ncls:                         # argmax over the first num_classes only (rare: host path)
        c = np.argmax(probs.cpu().numpy()[0, :num_classes], axis=0)
        colored = np.zeros((H, W, 3), dtype='uint8')
        for i in range(num_classes):
            colored[c == i, :] = COLORS[i + 1]
        return colored
    # class map -> colours on the device (the numpy masking loop was 2.6 of the call's 3.1 ms at 512^2)
    pal = _palette(device, eng.ncls)
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
    nv.call('iunet_colorize', nv.ptr(cls), H * W, nv.ptr(pal), min(num_classes, eng.ncls), nv.ptr(rgb), nv.stream())
    return rgb.cpu().numpy()


def predict_volume_array(model, volume, input_size=256, num_classes=2, overlap=0.25, batch_size=None,
                         axes=[0, 1, 2], block_range=None, accumulator=None, finalize=True):
    """predict.py:164-256 for one in-memory uint8 volume (numpy or device tensor): block grid,
    reflect-padded blocks, block prediction (2.5-D with a 2-D model, direct with a 3-D model),
    Gaussian blend, normalise + quantise.  Returns the uint8 [V, C] device tensor.
    `block_range=(lo, hi)` restricts to a contiguous run of the flat block list (multi-GPU)."""
    dev = model.device
    vol = volume if torch.is_tensor(volume) else torch.from_numpy(np.ascontiguousarray(volume))
    vol = vol.to(dev)
    V = tuple(vol.shape)
    S, C = int(input_size), int(num_classes)
    acc = accumulator or VolumeAccumulator(V, C, S, dev)
    bc, pbc, lbc = get_block_coordinates(np.array(V), input_size=S, overlap=overlap)
    lo, hi = (0, len(pbc)) if block_range is None else block_range
    eng = model.engine('eval')
    if eng.dim == 3:
        predict_blocks_3d(eng, acc, vol, bc, pbc, lbc, lo, hi)
    else:
        blk = torch.empty((S, S, S), dtype=torch.uint8, device=dev)
        for i in range(lo, hi):
            gather_block(vol, pbc[i], S, out=blk)
            predict_block_device(model, blk, acc.block_probs, C, batch_size, axes)
            acc.blend(bc[i], lbc[i])
    return acc.finalize() if finalize else acc


def predict_volumes(input_size=256, num_channels=1, num_classes=2, overlap=0.25, chunk_size=128, shard_size=256,
                    batch_size=None, axes=[0, 1, 2]):
    """predict.py:114-266: every data/image_volumes/<name>.zarr['0'] (uint8 [Z,Y,X]) -> data/predicted_volumes/<name>.zarr
    ['0'] uint8 [Z,Y,X,C], chunks (chunk_size,)*3 + (C,) inside shards (shard_size,)*3 + (C,) (predict.py:173-180), then the
    multiscale pyramid (predict.py:261).  The store is read and written by zarr3.py shard by shard through pinned staging;
    everything between -- block grid, reflect-padded blocks, block prediction, Gaussian blend, normalise + quantise, the
    pyramid levels -- stays in HBM (the reference keeps float32 accumulators in temporary Zarr arrays on disk)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import zarr3
    device = torch.device('cuda')
    model = _load_model(num_channels, num_classes, device)

    def store(final, done, save_path, name, shape, start):
        # encoding and writing the [Z, Y, X, C] result (zstd on the host cores, ~1 GB/s) takes about as long as predicting it: it runs
        # behind the next volume's read + prediction (one result in flight; its device tensor is this call's own).
        # The current device and stream are per THREAD: this worker starts on device 0 / the default stream whatever the caller had set.
        # It takes the result's device, runs its launches (the device-to-host copies, the pyramid's zoom kernels) on a stream of its
        # own, and makes that stream wait for the event the caller recorded behind the prediction.
        with torch.cuda.device(final.device):
            side = torch.cuda.Stream(device=final.device)
            side.wait_event(done)
            with torch.cuda.stream(side):
                root = zarr3.open(save_path, mode='w')
                arr = root.create_array(name='0', shape=list(final.shape), dtype='uint8', overwrite=True,
                                        chunks=(chunk_size,) * 3 + (num_classes,), shards=(shard_size,) * 3 + (num_classes,))
                arr.from_device(final)
                multiscale.add_multiscales(save_path, scale=0.5, level0=final)     # predict.py:261 (levels zoomed on the device)
                side.synchronize()                                               # `final` is released when this returns
        print(f'Completed volume {name} {shape} in {time.time() - start}.')

    pending = None
    with ThreadPoolExecutor(max_workers=1) as writer:
        for f in np.sort(glob.glob('data/image_volumes/*.zarr')):
            start = time.time()
            volume = zarr3.open(f, mode='r')['0'].to_device(device)
            eng = model.engine('eval')
            run = lambda: predict_volume_array(model, volume, input_size, num_classes, overlap, batch_size, axes)
            # split precision keeps act_scale x activation in fp16: the range flag covers every block of the volume; a volume whose
            # activations left that range is predicted again in a wider form (engine_auto.EngineAuto.run_checked) -- never a warning
            final = eng.run_checked(run) if hasattr(eng, 'run_checked') else run()
            if pending is not None:
                pending.result()                                           # (raises what the writer raised)
            done = torch.cuda.Event()
            done.record()                                                  # on the caller's current stream, behind the prediction
            pending = writer.submit(store, final, done, f.replace('image_volumes', 'predicted_volumes'), os.path.basename(f),
                                    tuple(volume.shape), start)
        if pending is not None:
            pending.result()
    print('\nAll volumes segmented.\n')
