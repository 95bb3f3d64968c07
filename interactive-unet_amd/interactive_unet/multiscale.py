"""Multiscale-pyramid part of the reference's interactive_unet/utils.py ("0.5x nearest
multiscale pyramid on device"): `resize_volume`, `add_multiscales`, `create_multiscale_zarr`, `read_volume` with the
reference's names, arguments and level arithmetic (utils.py:18-98).  It lives in a module the reference does NOT own
(`multiscale`), so that dropping the native modules into the reference's package leaves the reference's own `utils.py` --
project directories, TIFF import, plots, which app.py / annotator.py / volumedata.py call -- in place;
the package's `utils.py` re-exports these names for stand-alone use.  The zoom of every block runs on the GPU (libiunet:
iunet_zoom_nearest_u8, bit-exact with scipy.ndimage.zoom(block, scale, order=0) including its constant-fill of an
out-of-range last sample); volumes may be CUDA tensors (zoomed in place on the device) or host / Zarr arrays (each block
is staged through the device, as the reference stages it through scipy).  The rest of the reference's utils.py (project
directories, TIFF import, colours, plots) is outside the hot path.
"""
import itertools

import numpy as np
import torch

from . import _native as nv

_TABLES = {}


def _axis_table(n, scale, block_size):
    """Source index of every destination sample along one axis of resize_volume: the blocks [i0, i1) of `block_size` land
    in [int(i0*scale), int(i1*scale)), each with scipy's zoom table of its own length (-1 = constant 0).  Because
    ndimage.zoom is separable, the whole block loop equals ONE gather with these per-axis tables.  block_size None: the
    axis is not blocked (the channel axis of a 4-D volume)."""
    lib, parts, pos = nv.lib(), [], 0
    step = n if block_size is None else int(block_size)
    for i0 in range(0, n, step):
        i1 = min(i0 + step, n)
        t0, t1 = (0, int(lib.iunet_zoom_nearest_len(n, scale))) if block_size is None else (int(i0 * scale), int(i1 * scale))
        m = int(lib.iunet_zoom_nearest_len(i1 - i0, scale))
        if m != t1 - t0 or t0 != pos:       # numpy raises the same way when a zoomed block does not fit its slot
            raise ValueError(f'could not broadcast input array from shape ({m},) into shape ({t1 - t0},)')
        if m > 0:
            t = (nv.c_int * m)()
            nv.call('iunet_zoom_nearest_table', i1 - i0, scale, t, m)
            a = np.frombuffer(t, dtype=np.int32, count=m).copy()
            parts.append(np.where(a < 0, -1, a + i0).astype(np.int32))
        pos = t1
    return np.concatenate(parts) if parts else np.zeros(0, np.int32)


def _tables(shape, scale, block_size, device):
    """Device int32 tables (the four axes concatenated) of a whole resize_volume call and its destination shape; cached."""
    key = (tuple(int(n) for n in shape), float(scale), int(block_size), str(device))
    hit = _TABLES.get(key)
    if hit is None:
        parts = [_axis_table(n, key[1], key[2] if ax < 3 else None) for ax, n in enumerate(key[0])]
        tab = torch.from_numpy(np.concatenate(parts + [np.zeros(1, np.int32)])).to(device)
        hit = (tab, tuple(len(t) for t in parts))
        if len(_TABLES) > 64:
            _TABLES.clear()
        _TABLES[key] = hit
    return hit


def _resize_device(src, dst, scale, block_size):
    """The whole block loop of resize_volume as one gather launch: uint8 CUDA tensors of 3 or 4 axes."""
    nd = src.dim()
    if nd not in (3, 4) or src.dtype != torch.uint8 or dst.dtype != torch.uint8:
        raise ValueError(f'resize_volume on the device: uint8 volumes with 3 or 4 axes, got {tuple(src.shape)} {src.dtype}')
    tab, out_shape = _tables(src.shape, scale, block_size, src.device)
    # the reference writes dst_vol[int(i0*scale):int(i1*scale), ...] of the three blocked axes only: a larger destination keeps its
    # other samples; a smaller one (or a mismatching channel axis) is numpy's broadcast error
    if any(o > d for o, d in zip(out_shape[:3], dst.shape[:3])) or tuple(out_shape[3:]) != tuple(dst.shape[3:]):
        raise ValueError(f'could not broadcast input array from shape {out_shape} into shape {tuple(dst.shape)}')
    if 0 in out_shape:
        return
    view = dst[:out_shape[0], :out_shape[1], :out_shape[2]]
    dims = list(out_shape) + [1] * (4 - nd)
    sst = list(src.stride()) + [1] * (4 - nd)
    dstr = list(view.stride()) + [1] * (4 - nd)
    if dstr[3] != 1 or dstr[2] != dims[3]:
        raise ValueError('resize_volume on the device: the destination rows must be contiguous')
    if dims[0] > 65535 or dims[1] > 65535:
        raise ValueError('resize_volume on the device: at most 65535 samples along the two outer destination axes')
    sdims = [int(v) for v in src.shape] + [1] * (4 - nd)
    nv.call('iunet_zoom_nearest_u8', nv.ptr(src), (nv.c_int * 4)(*sdims), (nv.c_ll * 4)(*sst), nv.ptr(view), (nv.c_ll * 4)(*dstr),
            (nv.c_int * 4)(*dims), nv.ptr(tab), nv.stream())


def resize_volume(src_vol, dst_vol, scale=0.5, block_size=512, order=0):
    """utils.py:29-48.  Blocks of `block_size` along the first three axes; block (i0:i1, ...) lands in
    dst_vol[int(i0*scale):int(i1*scale), ...].  Two CUDA tensors: one launch for the whole loop.  Host / Zarr arrays on
    either side: the reference's block loop, each block staged through the device."""
    if order != 0:
        raise ValueError('resize_volume: only order=0 (nearest), the order the reference uses for its pyramids')
    dev_src = torch.is_tensor(src_vol) and src_vol.is_cuda
    dev_dst = torch.is_tensor(dst_vol) and dst_vol.is_cuda
    if dev_src and dev_dst:
        return _resize_device(src_vol, dst_vol, scale, int(block_size))
    device = src_vol.device if dev_src else (dst_vol.device if dev_dst else torch.device('cuda'))
    extent = [int(v) for v in src_vol.shape[:3]]
    starts = [range(0, n, block_size) for n in extent]
    for origin in itertools.product(*starts):
        src_box = tuple(slice(o, min(o + block_size, n)) for o, n in zip(origin, extent))
        dst_box = tuple(slice(int(b.start * scale), int(b.stop * scale)) for b in src_box)
        block = src_vol[src_box]
        if not dev_src:                              # host / Zarr source: the block travels to the device
            block = torch.from_numpy(np.ascontiguousarray(block)).to(device)
        if dev_dst:
            _resize_device(block, dst_vol[dst_box], scale, int(block_size))
        else:
            zoomed = tuple(b.stop - b.start for b in dst_box) + tuple(_tables(block.shape, scale, block_size, device)[1][3:])
            staged = torch.empty(zoomed, dtype=torch.uint8, device=device)
            _resize_device(block, staged, scale, int(block_size))
            dst_vol[dst_box] = staged.cpu().numpy()


def num_multiscale_steps(volume_shape, chunk_shape, scale=0.5):
    """utils.py:59-60: number of downscale steps until the volume fits inside a chunk."""
    return int(np.floor(np.log((np.array(volume_shape) / np.array(chunk_shape)).max()) / np.log(1 / scale)))


def multiscale_levels(volume, chunk_shape, shard_shape, scale=0.5):
    """The array side of add_multiscales (utils.py:50-77): levels 1 .. n of a uint8 CUDA volume (level 0), each
    resize_volume(previous, scale, block_size=shard_shape[0])."""
    levels, z0 = [], volume
    for _ in range(num_multiscale_steps(tuple(volume.shape), chunk_shape, scale)):
        shape1 = tuple(int(x * scale) for x in z0.shape)
        if 0 in shape1:
            # utils.py:64 scales EVERY axis, the class axis of a prediction too: 2 classes -> 1 -> 0.  The reference's zoom of an
            # empty array raises (volumes of >= 8 chunks per axis, i.e. 1024^3 at 128^3 chunks); here the pyramid ends at the last
            # level that exists, after level 0 and the levels before it have been written
            break
        z1 = torch.empty(shape1, dtype=z0.dtype, device=z0.device)
        resize_volume(z0, z1, scale=scale, block_size=int(shard_shape[0]), order=0)
        levels.append(z1)
        z0 = z1
    return levels


def read_volume(path, level=0):
    """utils.py:18-27 (including its clip of `level` to the NUMBER of scales, one past the last level)."""
    from . import zarr3
    root = zarr3.open(path, mode='r')
    num_scales = len(np.sort(list(root.array_keys())))
    level = int(np.clip(level, 0, num_scales))
    return root[str(level)]


def add_multiscales(src_file, scale=0.5, level0=None):
    """utils.py:50-77: levels '1' .. 'n' beside level '0' of a Zarr group, same chunks / shards / dtype.  Level 0 goes
    to the device once (shard by shard through pinned staging; or pass the resident tensor as `level0`), every level is
    one gather launch there (multiscale_levels), and each level is written back shard by shard."""
    from . import zarr3
    root = zarr3.open(src_file, mode='r+')
    a0 = root['0']
    chunk_shape, shard_shape = a0.chunks, a0.shards or a0.chunks
    vol = level0 if level0 is not None else a0.to_device('cuda')
    for i, lv in enumerate(multiscale_levels(vol, chunk_shape, shard_shape, scale)):
        z1 = root.create_array(name=str(i + 1), shape=tuple(lv.shape), chunks=chunk_shape, shards=a0.shards, dtype=a0.dtype,
                               overwrite=True)
        z1.from_device(lv)


def create_multiscale_zarr(volume, dst_file, scale=0.5, chunk_size=128, shard_size=256):
    """utils.py:79-98: level '0' = `volume` (numpy array or tensor), chunks chunk_size^3 inside shards shard_size^3, then
    the pyramid."""
    from . import zarr3
    chunk_shape, shard_shape = (chunk_size,) * 3, (shard_size,) * 3
    root = zarr3.open(dst_file, mode='w')
    vol = volume if torch.is_tensor(volume) else torch.from_numpy(np.ascontiguousarray(volume))
    z0 = root.create_array(name='0', shape=tuple(vol.shape), chunks=chunk_shape, shards=shard_shape,
                           dtype=str(vol.dtype).replace('torch.', ''), overwrite=True)
    z0.from_device(vol)
    add_multiscales(dst_file, scale=scale, level0=vol if vol.is_cuda else None)
