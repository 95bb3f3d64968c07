"""Oracle vs golden vectors captured from the reference (tests/golden/make_golden.py).
CPU only.  Pins oracle/{metrics,predict,slicer,multiscale,loader}_ref.py to the reference's own outputs."""
import os
import numpy as np
import pytest

from oracle import metrics_ref, predict_ref, slicer_ref


@pytest.fixture(scope='module')
def G(golden_dir):
    return {k: np.load(os.path.join(golden_dir, k + '.npz')) for k in ('losses', 'predict', 'slicer')}


def test_losses_and_grads_match_reference(G):
    g = G['losses']
    for c in range(int(g['n_cases'])):
        key = f'c{c}'
        p, y, axes = g[key + '_p'], g[key + '_y'], list(g[key + '_axes'])
        w = g[key + '_w'] if key + '_w' in g else None
        for kind in metrics_ref.KINDS:
            want = float(g[f'{key}_{kind}'])
            got = metrics_ref.loss(kind, p, y, w, axes)
            assert abs(got - want) <= 1e-10 * max(1, abs(want)), (c, kind, got, want)
            gw = g[f'{key}_{kind}_grad']
            gg = metrics_ref.loss_grad(kind, p, y, w, axes)
            assert np.allclose(gg, gw, rtol=1e-8, atol=1e-12), (c, kind, np.abs(gg - gw).max())
        r = metrics_ref.rounded_metrics(p, y, w, axes)
        assert np.allclose(r, g[key + '_rounded'], rtol=1e-10, atol=1e-12)


def test_block_coordinates_bit_exact(G):
    g = G['predict']
    for k in range(int(g['n_bc'])):
        a = g[f'bc{k}_args']
        b, pb, lb = predict_ref.get_block_coordinates(a[:3], int(a[3]), a[4] / 100.0)
        assert np.array_equal(b, g[f'bc{k}_b']) and np.array_equal(pb, g[f'bc{k}_pb']) \
            and np.array_equal(lb, g[f'bc{k}_lb']), a


def test_c4_block_grid():
    b, pb, lb = predict_ref.get_block_coordinates((1024,) * 3, 128, 0.25)
    assert len(pb) == 1331 and tuple(pb[0][:3]) == (-32,) * 3 and tuple(pb[-1][3:]) == (1056,) * 3


def test_padded_block_and_reflect(G):
    g = G['predict']
    vol = g['pad_vol']
    for i, c in enumerate(g['pad_coords']):
        assert np.array_equal(predict_ref.get_padded_block(vol, *c), g[f'pad{i}'])
    for n in (1, 2, 7, 16):
        assert np.array_equal(predict_ref.reflect_index(g['reflect_idx'], n), g[f'reflect_{n}'])


def test_windows(G):
    g = G['predict']
    for S in (8, 16, 32):
        assert np.array_equal(predict_ref.gaussian_3d(S), g[f'gauss{S}'])
        assert np.allclose(predict_ref.hanning_3d(S), g[f'hann{S}'], rtol=0, atol=0)
    w = predict_ref.gaussian_3d(128)
    assert np.array_equal(np.array([w[i, i, i] for i in range(128)]), g['gauss128_diag'])
    assert np.array_equal(w[64, 64, :], g['gauss128_line'])
    assert np.array_equal(predict_ref.get_shard_coordinates([300, 260, 129], 128), g['shards_300_260_129_128'])
    assert np.array_equal(predict_ref.get_shard_coordinates([512] * 3, 256), g['shards_512_256'])


def _stub(kind):
    def softmax(l):
        e = np.exp(l - l.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)

    def f(x):
        x = x.astype(np.float32)
        B, _, H, W = x.shape
        if kind == 0:
            return softmax(np.concatenate([x, 1 - x], 1))
        r = (np.arange(H, dtype=np.float32) / H).reshape(1, 1, H, 1)
        c = (np.arange(W, dtype=np.float32) / W).reshape(1, 1, 1, W)
        return softmax(np.concatenate([x * (1 + r), x * (0.5 + 2 * c) - 0.3 * r, 0.2 + 0 * x], 1))
    return f


def test_predict_block_2p5d(G):
    g = G['predict']
    for k in range(int(g['n_pb'])):
        S, bs, kind, ncls = [int(v) for v in g[f'pb{k}_args']]
        out = predict_ref.predict_block(_stub(kind), g[f'pb{k}_block'], ncls, bs, list(g[f'pb{k}_axes']))
        assert np.allclose(out, g[f'pb{k}_out'], rtol=0, atol=2e-6), np.abs(out - g[f'pb{k}_out']).max()


def test_blend_normalise_quantise(G):
    g = G['predict']
    fn = lambda blk: predict_ref.predict_block(_stub(1), blk, 3, 8, (0, 1, 2))
    final, pred, weight = predict_ref.blend_volume(g['blend_volume'], fn, 32, 3, 0.25)
    assert np.allclose(weight, g['blend_weight'], rtol=1e-6, atol=0)
    assert np.allclose(pred[::7, ::5, ::3], g['blend_pred_sample'], rtol=0, atol=5e-6)
    diff = np.abs(final.astype(int) - g['blend_final'].astype(int))
    # truncating cast: a 1e-6 float wobble in the stub softmax may flip a value sitting on an integer
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


def test_slicer_geometry(G):
    g = G['slicer']
    ramp, vol = g['ramp'], g['vol']
    for i in range(int(g['n'])):
        rot_vec, R, u, v, w = slicer_ref.orientation_vectors(g[f's{i}_rv'])
        for name, val in (('rotvec', rot_vec), ('rotmat', R), ('u', u), ('v', v), ('w', w)):
            assert np.array_equal(val, g[f's{i}_{name}']), (i, name)
        origin = g[f's{i}_origin']
        assert np.array_equal(slicer_ref.interpolation_coords(u, v, w, origin, 8), g[f's{i}_coords8'])
        for axis in (0, 1, 2):
            for order in (0, 1):
                s = slicer_ref.get_slice(ramp, u, v, w, origin, axis, 24, order)
                assert np.array_equal(s, g[f's{i}_slice_a{axis}_o{order}']), (i, axis, order)
        upd = slicer_ref.update_volume(g[f's{i}_upd_data'], vol.copy(), u, v, w, origin, axis=1)
        assert np.array_equal(upd, g[f's{i}_upd_vol'])
        # to_dict/from_dict re-derives the vectors from the *normalised* rotation vector (slicer.py:84-92)
        assert np.array_equal(slicer_ref.orientation_vectors(rot_vec)[2], g[f's{i}_rt_u'])


def test_e4m3_rounding_matches_torch_float8():
    """oracle.unet_ref.round_e4m3 (the definition the device quantiser is tested against) equals torch's
    float8_e4m3fn conversion on the whole finite range, including subnormals and ties."""
    import numpy as np
    import torch
    from oracle import unet_ref
    rng = np.random.default_rng(0)
    x = np.concatenate([(rng.standard_normal(200000) * s).astype(np.float32) for s in (100, 1, 0.01)]).clip(-448, 448)
    ties = np.float32([0, 448, -448, 2 ** -9, 2 ** -10, 3 * 2 ** -10, 2 ** -6, 1.0625, 1.1875, 17, 19, 416, 432, 447.9])
    x = np.concatenate([x, ties, -ties])
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(unet_ref.round_e4m3(x), want)
    w = torch.randn(16, 8, 3, 3, 3, generator=torch.Generator().manual_seed(1)) * 0.07
    q = unet_ref.quantize_e4m3(w)
    assert torch.equal(q.half().float(), q) and torch.equal(q.bfloat16().float(), q)        # exact in both 16-bit types
    assert (q - w).abs().max() <= w.abs().amax(dim=(1, 2, 3, 4)).max() * 2 ** -4             # half an e4m3 step at the top binade


def test_multiscale_oracle_matches_reference_resize_volume(golden_dir):
    """oracle/multiscale_ref.py against the reference's resize_volume (utils.py:29-48) on the golden volumes: several
    blocks per axis, ragged last blocks, the sizes whose last sample scipy fills with 0, 4-D prediction volumes."""
    from oracle import multiscale_ref as mr
    g = np.load(os.path.join(golden_dir, 'multiscale.npz'))
    zeros = 0
    for i in range(int(g['n'])):
        src, want, block = g[f'c{i}_src'], g[f'c{i}_dst'], int(g[f'c{i}_block'])
        got = np.full_like(want, 7)
        mr.resize_volume(src, got, 0.5, block)
        assert np.array_equal(got, want), i
        zeros += int((want == 0).sum())
    assert zeros > 0          # the constant-fill quirk is exercised (the sources hold no zeros)


def test_multiscale_oracle_matches_scipy_zoom():
    """The per-axis index table against scipy.ndimage.zoom(order=0) itself (the third-party call under utils.py:46) for every
    length 1..300, powers of two and their neighbours, several zoom factors; N-d zoom on random uint8 blocks."""
    from scipy import ndimage
    from oracle import multiscale_ref as mr
    for n in list(range(1, 301)) + [511, 512, 513, 1000, 1023, 1024, 1025, 2048]:
        a = np.arange(1, n + 1, dtype=np.int64)
        for zoom in (0.5, 0.25, 0.3, 0.75, 0.125):
            t = mr.zoom_table(n, zoom)
            if len(t) == 0:
                continue
            assert np.array_equal(ndimage.zoom(a, zoom, order=0), np.where(t < 0, 0, a[np.clip(t, 0, n - 1)])), (n, zoom)
    rng = np.random.default_rng(0)
    for shape in [(40, 33, 70), (32, 48, 56), (7, 9, 11), (20, 20, 20, 2), (16, 16, 16, 3)]:
        x = rng.integers(1, 255, shape, dtype=np.uint8)
        for zoom in (0.5, 0.25):
            assert np.array_equal(ndimage.zoom(x, zoom, order=0), mr.zoom_nearest(x, zoom)), (shape, zoom)
    assert mr.num_steps((1024,) * 3, (128,) * 3) == 3 and mr.num_steps((300, 260, 129), (128,) * 3) == 1


def test_loader_normalisation_matches_reference(golden_dir):
    """oracle/loader_ref.normalise against the reference's own normalisation block (loader.py:32-42): /255 through float64,
    weight repeated over the classes, mask and weight zeroed where image channel 0 is 0."""
    from oracle import loader_ref as lr
    g = np.load(os.path.join(golden_dir, 'loader.npz'))
    for k in range(int(g['n'])):
        got = lr.normalise(g[f'c{k}_image'], g[f'c{k}_mask'], g[f'c{k}_weight'])
        for x, name in zip(got, ('image', 'mask', 'weight')):
            want = g[f'c{k}_{name}_f']
            assert x.dtype == want.dtype and np.array_equal(x, want), (k, name)


def test_loader_index_arithmetic_matches_torch_primitives():
    """The explicit per-pixel index arithmetic of the transform chain (what the HIP kernel implements) against the same chain
    run through torch's own flip / grid_sample(nearest, zeros, align_corners=False) / interpolate(nearest): identical on every
    pixel -- random angles, the rot90 fast paths, square and non-square annotations, crops from torchvision's parameter rule."""
    import torch
    from oracle import loader_ref as lr
    rng = np.random.default_rng(0)
    fixed = [0.0, 180.0, 90.0, -90.0, 360.0, -180.0, 270.0, 45.0]
    for trial in range(24):
        H, W = [(512, 512), (300, 400), (256, 256), (200, 150), (129, 333)][trial % 5]
        x = rng.random((2, H, W)).astype(np.float32)
        hf, vf = bool(rng.integers(2)), bool(rng.integers(2))
        ang = fixed[trial] if trial < len(fixed) else float(rng.uniform(-360, 360))
        crop = lr.resized_crop_params(H, W, lambda a, b: float(rng.uniform(a, b)), lambda n: int(rng.integers(n)))
        want = lr.transform_reference_ops(x, hf, vf, ang, crop).numpy()
        got = lr.transform(x, hf, vf, ang, crop)
        assert np.array_equal(got, want), (trial, H, W, ang, crop)
    i, j, h, w = lr.resized_crop_params(100, 1000, lambda a, b: 1.0 if a < 0.5 else b, lambda n: 0)      # never fits: centre-crop fallback
    assert (h, w) == (100, 133) and (i, j) == (0, 433)
