"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the reference's
multiscale pyramid for uint8 volumes.

Follows
  /root/reference/interactive_unet/utils.py:29-48   resize_volume   (block loop; dst[int(i0*s):int(i1*s)] = zoom(src[i0:i1]))
  /root/reference/interactive_unet/utils.py:50-77   add_multiscales (number of levels, level shapes)
and the published algorithm of the third-party call in it, scipy.ndimage.zoom(x, zoom, order=0) (scipy 1.15.3,
_interpolation.py zoom + ni_interpolation.c NI_ZoomShift; mode 'constant', cval 0, grid_mode False): output length
round(n * zoom) (Python round: half to even), per-axis coordinate cc = o * (n_in - 1) / (n_out - 1) in double, sample
floor(cc + 0.5), cval where cc > n_in - 1.

Pinned by tests/test_oracle_golden.py against scipy.ndimage.zoom itself (present in the image) and against
tests/golden/multiscale.npz, produced by the reference's own resize_volume (tests/golden/make_golden.py).
"""
import math
import numpy as np


def zoom_len(n_in, zoom):
    return int(round(n_in * zoom))


def zoom_table(n_in, zoom):
    """Per-axis source index of every output sample; -1 = scipy's constant (0)."""
    n_out = zoom_len(n_in, zoom)
    z = (n_in - 1) / (n_out - 1) if n_out - 1 > 0 else 1.0
    t = np.empty(n_out, dtype=np.int64)
    for o in range(n_out):
        cc = o * z
        t[o] = -1 if (cc < 0 or cc > n_in - 1) else int(math.floor(cc + 0.5))
    return t


def zoom_nearest(x, zoom):
    """ndimage.zoom(x, zoom, order=0) for an N-d integer array (every axis zoomed, as the reference calls it)."""
    x = np.asarray(x)
    tabs = [zoom_table(n, zoom) for n in x.shape]
    out = x
    for ax, t in enumerate(tabs):
        out = np.take(out, np.clip(t, 0, None), axis=ax)
        if (t < 0).any():
            sl = [slice(None)] * out.ndim
            sl[ax] = np.nonzero(t < 0)[0]
            out = out.copy()
            out[tuple(sl)] = 0
    return out


def resize_volume(src_vol, dst_vol, scale=0.5, block_size=512):
    """utils.py:29-48 (order 0): blocks of block_size along the first three axes."""
    n = src_vol.shape
    for i in range(0, n[0], block_size):
        i0, i1 = i, min(i + block_size, n[0])
        for j in range(0, n[1], block_size):
            j0, j1 = j, min(j + block_size, n[1])
            for k in range(0, n[2], block_size):
                k0, k1 = k, min(k + block_size, n[2])
                dst_vol[int(i0 * scale):int(i1 * scale), int(j0 * scale):int(j1 * scale), int(k0 * scale):int(k1 * scale)] = \
                    zoom_nearest(src_vol[i0:i1, j0:j1, k0:k1], scale)


def num_steps(volume_shape, chunk_shape, scale=0.5):
    """utils.py:59-60: downscale steps until the volume fits inside a chunk."""
    return int(np.floor(np.log((np.array(volume_shape) / np.array(chunk_shape)).max()) / np.log(1 / scale)))


def multiscale_levels(volume, chunk_shape, shard_shape, scale=0.5):
    """utils.py:50-77 on arrays: the list of levels 1 .. num_steps of `volume` (level 0)."""
    levels, z0 = [], np.asarray(volume)
    for _ in range(num_steps(z0.shape, chunk_shape, scale)):
        z1 = np.zeros(tuple(int(x * scale) for x in z0.shape), dtype=z0.dtype)
        resize_volume(z0, z1, scale=scale, block_size=shard_shape[0])
        levels.append(z1)
        z0 = z1
    return levels
