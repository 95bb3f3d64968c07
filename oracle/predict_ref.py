"""Oracle: restatement of the reference's volume-prediction numerics (numpy).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Each function cites the lines of
/root/reference/interactive_unet/predict.py it follows.  Integer results (block
coordinates, padded blocks, quantised uint8 output) must match the goldens bit for
bit; float32 results (windows, blended probabilities) to float32 rounding.
"""
import math
import numpy as np


def get_block_coordinates(volume_shape, input_size=256, overlap=0.25):
    """predict.py:362-411.  Returns (block, padded_block, local) int arrays [n,6].

    n/axis = ceil((V - o*S) / (S - o*S)); padded extent = round(n*S - (n-1)*S*o);
    shift = (padded - V) // 2; padded start k = int(k*S*(1-o) - shift) (truncation
    toward zero of a float, predict.py:390-391)."""
    V = np.asarray(volume_shape, dtype=np.int64)
    S, o = input_size, overlap
    n = np.ceil((V - o * S) / (S - o * S)).astype(int)
    padded = np.round(n * S - (n - 1) * S * o).astype(int)
    shift = (padded - V) // 2
    blocks, pblocks, locs = [], [], []
    for i in range(n[0]):
        for j in range(n[1]):
            for k in range(n[2]):
                lo = np.array([i, j, k], dtype=np.float64) * S * (1 - o)
                c = np.concatenate([lo, lo + S]) - np.concatenate([shift, shift])
                c = c.astype(int)                       # float -> int truncation
                pblocks.append(c)
                lo_c = np.maximum(c[:3], 0)
                hi_c = np.minimum(c[3:], V)
                blocks.append(np.concatenate([lo_c, hi_c]))
                locs.append(np.concatenate([lo_c - c[:3], hi_c - c[:3]]))
    return np.array(blocks), np.array(pblocks), np.array(locs)


def reflect_index(idx, size):
    """predict.py:270-279: mirror without repeating the edge, period 2*size-2."""
    idx = np.asarray(idx)
    if size == 1:
        return np.zeros_like(idx)
    period = 2 * size - 2
    idx = np.abs(idx) % period
    return np.where(idx < size, idx, period - idx)


def get_padded_block(volume, i0, j0, k0, i1, j1, k1):
    """predict.py:291-316 (the live, second definition): clip, read, np.pad(reflect).

    Restated as an index gather; identical to np.pad(mode='reflect') whenever each pad
    width is smaller than the clipped extent along that axis (always true for the
    reference's block grid when V >= S/2 ... np.pad re-reflects otherwise)."""
    V = volume.shape
    lo = [max(i0, 0), max(j0, 0), max(k0, 0)]
    hi = [min(i1, V[0]), min(j1, V[1]), min(k1, V[2])]
    block = np.asarray(volume[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]])
    idx = []
    for a, (p0, p1) in enumerate(((i0, i1), (j0, j1), (k0, k1))):
        n = hi[a] - lo[a]
        rel = np.arange(p0, p1) - lo[a]
        # np.pad 'reflect' keeps reflecting for pads wider than the data
        idx.append(reflect_index(rel, n))
    return block[np.ix_(*idx)]


def get_shard_coordinates(volume_shape, shard_size=128):
    """predict.py:318-325."""
    V = np.asarray(volume_shape)
    starts = [np.arange(0, s, shard_size) for s in V]
    c = np.stack(np.meshgrid(*starts, indexing='ij'), -1).reshape(-1, 3)
    return np.concatenate([c, np.minimum(c + shard_size, V)], axis=1)


def gaussian_3d(input_size, sigma=0.125, eps=1e-3):
    """predict.py:327-347.  float32 throughout, like the reference (np.float32 coords,
    float32 exp, float32 outer product)."""
    sigma = sigma * input_size
    coords = np.arange(input_size, dtype=np.float32) - (input_size - 1) / 2.0
    g = np.exp(-(coords ** 2) / (2 * sigma ** 2)).astype(np.float32)
    g /= g.max()
    w = g[:, None, None] * g[None, :, None] * g[None, None, :]
    w /= w.max()
    return np.clip(w, max(w.min(), eps), 1.0)


def hanning_3d(input_size, eps=1e-3):
    """predict.py:349-360."""
    h = np.hanning(input_size)
    w = h[:, None, None] * h[None, :, None] * h[None, None, :]
    w /= w.max()
    return np.clip(w, max(w.min(), eps), 1.0).astype('float32')


def predict_block(model_fn, block, num_classes=2, batch_size=8, axes=(0, 1, 2)):
    """predict.py:79-112, 2.5-D prediction.  `model_fn(batch[B,1,S,S] float32) ->
    probs[B,C,S,S]` (numpy).  For each axis the block is viewed with that axis first,
    the 2-D net is run on batches of slices and the probabilities are added back in
    the block's own orientation; the sum is divided by len(axes)."""
    S = block.shape[0]
    out = np.zeros((S, S, S, num_classes), dtype=np.float32)
    for axis in axes:
        view = np.moveaxis(block, axis, 0)
        for i in range(0, S, batch_size):
            p = model_fn(np.ascontiguousarray(view[i:i + batch_size])[:, None])
            p = np.transpose(p, (0, 2, 3, 1)).astype(np.float32)      # B,H,W,C
            if axis == 0:
                out[i:i + batch_size] += p
            elif axis == 1:
                out[:, i:i + batch_size] += p.transpose(1, 0, 2, 3)
            else:
                out[:, :, i:i + batch_size] += p.transpose(1, 2, 0, 3)
    out /= len(axes)
    return out


def blend_volume(volume_u8, block_fn, input_size, num_classes, overlap=0.25,
                 window=None):
    """predict.py:201-256 on an in-memory volume: for every block
    pred += P * win, weight += win (predict.py:244-245), then
    uint8(255 * pred / max(weight, 1e-3)) with a truncating cast (predict.py:255).
    `block_fn(block float32 [S,S,S] in [0,1]) -> probs float32 [S,S,S,C]`."""
    V = np.array(volume_u8.shape)
    if window is None:
        window = gaussian_3d(input_size, 0.125)
    pred = np.zeros(tuple(V) + (num_classes,), np.float32)
    weight = np.zeros(tuple(V), np.float32)
    bc, pbc, lbc = get_block_coordinates(V, input_size, overlap)
    for b, pb, lb in zip(bc, pbc, lbc):
        blk = get_padded_block(volume_u8, *pb).astype('float32') / 255.0
        P = block_fn(blk)
        i0, j0, k0, i1, j1, k1 = b
        a0, b0, c0, a1, b1, c1 = lb
        pred[i0:i1, j0:j1, k0:k1] += P[a0:a1, b0:b1, c0:c1, :] * window[a0:a1, b0:b1, c0:c1, None]
        weight[i0:i1, j0:j1, k0:k1] += window[a0:a1, b0:b1, c0:c1]
    final = (255 * pred / np.maximum(weight, 1e-3)[..., None]).astype('uint8')
    return final, pred, weight


def predict_slice_post(y_prob_nchw, num_classes):
    """predict.py:37-42 post-processing: argmax over the first num_classes channels,
    one-hot * 255 uint8, palette colours utils.py:304-306, :351-357."""
    COLORS = np.array([[0, 0, 0], [230, 25, 75], [60, 180, 75], [255, 225, 25],
                       [0, 130, 200], [245, 130, 48], [145, 30, 180], [70, 240, 240],
                       [240, 50, 230], [210, 245, 60], [170, 255, 195]], dtype=np.uint8)
    y = np.moveaxis(y_prob_nchw, 1, -1)
    cls = np.argmax(y[0, :, :, :num_classes], axis=-1)
    onehot = (np.stack([cls == i for i in range(num_classes)], -1) * 255).astype('uint8')
    colored = np.zeros(cls.shape + (3,), dtype='uint8')
    for i in range(num_classes):
        colored[onehot[:, :, i] == 255, :] = COLORS[i + 1]
    return cls, onehot, colored
