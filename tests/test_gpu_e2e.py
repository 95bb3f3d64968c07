"""End-to-end GPU tests of the drop-in entry points: whole-volume prediction against the oracle's
restatement of predict.py:201-256, the sharded path on one rank, and trainer.train_model's files."""
import csv
import glob
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref, predict_ref


def _model(dim, ncls=2, seed=3, dtype='fp16'):
    from interactive_unet.unet import UNet
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=ncls, dim=dim, act_dtype=dtype, pretrained=False)
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=seed, randomize_bn=True)
    m.load_named(p)
    return m.cuda().eval(), p


def _volume(shape, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), 2.5)
    return (255 * (v - v.min()) / (v.max() - v.min())).astype(np.uint8)


@pytest.mark.parametrize('dim', [2, 3])
def test_predict_volume_array_vs_oracle(dim):
    """Block grid + reflect blocks + (2.5-D | 3-D) block prediction + Gaussian blend + truncating
    quantisation on device vs the oracle pipeline fed by the oracle network (same rounding points)."""
    from interactive_unet import predict
    S, C = 32, 2
    V = (72, 56, 40)
    model, p = _model(dim, C)
    vol = _volume(V, 21)
    got = predict.predict_volume_array(model, vol, input_size=S, num_classes=C, overlap=0.25).cpu().numpy()

    if dim == 2:
        net2d = lambda b: unet_ref.forward(p, torch.tensor(b), dim=2, act_dtype=torch.float16).numpy()
        block_fn = lambda blk: predict_ref.predict_block(net2d, blk, C, 8, (0, 1, 2))
    else:
        def block_fn(blk):
            pr = unet_ref.forward(p, torch.tensor(blk)[None, None], dim=3, act_dtype=torch.float16)
            return pr[0].permute(1, 2, 3, 0).contiguous().numpy()
    want, _, _ = predict_ref.blend_volume(vol, block_fn, S, C, 0.25)
    d = np.abs(got.astype(int) - want.astype(int))
    print(f'{dim}-D volume predict: max |uint8 diff| = {d.max()}, differing = {(d > 0).mean():.4f}')
    # probabilities carry the 16-bit-storage noise of the network (<= 3e-3): 255 * 3e-3 < 1 LSB,
    # plus the truncating cast -> at most 2 LSB anywhere, and the class map equal where classes differ by > 2 LSB
    assert d.max() <= 2
    sure = np.abs(want[..., 0].astype(int) - want[..., 1].astype(int)) > 4
    assert np.array_equal(got.argmax(-1)[sure], want.argmax(-1)[sure])


def test_sharded_predict_single_rank_equals_unsharded():
    from interactive_unet import predict, shard
    S, C = 32, 2
    V = (40, 40, 40)
    model, _ = _model(3, C)
    vol = torch.tensor(_volume(V, 22)).cuda()
    ref = predict.predict_volume_array(model, vol, input_size=S, num_classes=C).cpu().numpy()
    ops = shard.NativeOps(model, C, S)
    out, stats = shard.predict_volume_sharded(ops, vol, V, S, 0.25)
    assert stats['blocks'] == len(predict.get_block_coordinates(np.array(V), S, 0.25)[0])
    assert np.array_equal(out.cpu().numpy(), ref)


def test_predict_slice_vs_oracle():
    """predict.py:16-47 against the ORACLE: the colours are the palette (utils.py:304-306) applied to the argmax of the
    oracle network's probabilities (oracle/predict_ref.predict_slice_post) on every pixel whose oracle top-2 margin is
    beyond the 16-bit storage noise, and the returned probabilities are the oracle's; in the fp32 parity mode the whole
    coloured image is identical."""
    from interactive_unet import predict
    C = 3
    img = _volume((64, 96), 23)
    x = torch.tensor(img)[None, None].float() / 255.0
    for dtype, tol in (('fp16', 3e-3), ('fp32', 1e-5)):
        model, p = _model(2, C, dtype=dtype)
        want_p = unet_ref.forward(p, x, dim=2).numpy()                       # fp32 oracle, NCHW
        cls, onehot, want_col = predict_ref.predict_slice_post(want_p, C)
        colored = predict.predict_slice(img, num_classes=C, model=model)
        assert colored.shape == (64, 96, 3) and colored.dtype == np.uint8
        probs = predict.predict_slice(img, num_classes=C, return_probabilities=True, model=model)
        assert probs.shape == (1, 64, 96, C)
        err = np.abs(probs - np.moveaxis(want_p, 1, -1)).max()
        top2 = np.sort(want_p[0], axis=0)
        sure = (top2[-1] - top2[-2]) > 2 * max(err, 1e-6)
        print(f'predict_slice {dtype}: max |prob - oracle| = {err:.2e}, sure pixels {sure.mean():.3f}')
        assert err <= tol * 4 and sure.mean() > 0.9
        assert np.array_equal(colored[sure], want_col[sure])
        if dtype == 'fp32':
            assert np.array_equal(colored, want_col)


def test_find_max_batch_size_contract():
    """predict.py:49-77's contract without the OOM probing: a power-of-two multiple of `start`, at least `start`, at most
    `max_limit`, non-increasing in the slice size, and a batch of that size really runs."""
    from interactive_unet import predict
    model, _ = _model(2, 2)
    sizes = [predict.find_max_batch_size(model, input_size=s, start=4, max_limit=512) for s in (128, 256, 512)]
    for b in sizes:
        assert 4 <= b <= 512 and (b // 4) & (b // 4 - 1) == 0 and b % 4 == 0
    assert sizes[0] >= sizes[1] >= sizes[2]
    assert predict.find_max_batch_size(model, input_size=128, start=4, max_limit=8) == 8
    b = min(sizes[0], 64)
    x = torch.zeros((b, 1, 128, 128), dtype=torch.uint8, device='cuda')
    assert model(x).shape == (b, 2, 128, 128)


def test_predict_volumes_zarr_end_to_end(tmp_path, monkeypatch):
    """predict.py:114-266 through the store: data/image_volumes/<name>.zarr (written by utils.create_multiscale_zarr's
    layout: chunks inside shards) -> data/predicted_volumes/<name>.zarr['0'] uint8 [Z,Y,X,C] + pyramid levels, read and
    written by the built-in Zarr v3 reader / writer (zarr3.py) shard by shard through pinned staging."""
    from interactive_unet import predict, multiscale, zarr3
    monkeypatch.chdir(tmp_path)
    S, C, V = 32, 2, (72, 56, 40)
    model, _ = _model(3, C)
    os.makedirs('model')
    model.save_checkpoint(os.path.join('model', 'model.ckpt'))               # predict.py:22-24 loads it
    vol = _volume(V, 41)
    os.makedirs(os.path.join('data', 'image_volumes'))
    os.makedirs(os.path.join('data', 'predicted_volumes'))
    multiscale.create_multiscale_zarr(vol, os.path.join('data', 'image_volumes', 'a.zarr'), chunk_size=16, shard_size=32)
    src = zarr3.open(os.path.join('data', 'image_volumes', 'a.zarr'))
    assert np.array_equal(src['0'][...], vol) and src['0'].chunks == (16, 16, 16) and src['0'].shards == (32, 32, 32)
    assert src.array_keys() == ['0', '1', '2']                               # 72 -> 36 -> 18 (fits a 16^3 chunk at 2 steps)
    # a second volume of another shape: its read + prediction overlap the first one's encode + write (predict_volumes' write-behind)
    vol_b = _volume((40, 64, 48), 43)
    multiscale.create_multiscale_zarr(vol_b, os.path.join('data', 'image_volumes', 'b.zarr'), chunk_size=16, shard_size=32)
    wants = {'a': predict.predict_volume_array(model, vol, input_size=S, num_classes=C).cpu().numpy(),
             'b': predict.predict_volume_array(model, vol_b, input_size=S, num_classes=C).cpu().numpy()}
    predict.predict_volumes(input_size=S, num_classes=C, chunk_size=16, shard_size=32)
    for name, want in wants.items():
        out = zarr3.open(os.path.join('data', 'predicted_volumes', f'{name}.zarr'))
        a0 = out['0']
        assert a0.shape == want.shape and a0.chunks == (16, 16, 16, C) and a0.shards == (32, 32, 32, C)
        assert np.array_equal(a0[...], want)
        levels = multiscale.multiscale_levels(torch.tensor(want).cuda(), a0.chunks, a0.shards)
        assert out.array_keys() == [str(i) for i in range(len(levels) + 1)]
        for i, lv in enumerate(levels):
            assert np.array_equal(out[str(i + 1)][...], lv.cpu().numpy())
        assert multiscale.read_volume(os.path.join('data', 'predicted_volumes', f'{name}.zarr'), level=1).shape == tuple(levels[0].shape)


def test_predict_volumes_reruns_a_saturated_volume_in_a_wider_form(tmp_path, monkeypatch, capsys):
    """predict.py:114-266 with a checkpoint of the DEFAULT module (split-precision prediction) whose first activation is ~3000: beyond the
    fp16 range of the default form.  The volume must come out as the fp32 mode predicts it (within the truncation's 1 LSB), after the
    automatic second pass in the wider form -- never a warning beside a wrong result."""
    from interactive_unet import predict, multiscale, zarr3
    from interactive_unet.unet import UNet
    monkeypatch.chdir(tmp_path)
    S, C, V = 32, 2, (40, 56, 40)
    gain = 3000.0
    p = unet_ref.init_params(dim=3, ncls=C, seed=4, randomize_bn=True)
    p['enc0.conv1.weight'] = p['enc0.conv1.weight'] * gain
    p['enc0.bn1.bias'] = p['enc0.bn1.bias'] * gain
    p['enc0.bn1.running_mean'] = p['enc0.bn1.running_mean'] * gain
    p['enc0.conv2.weight'] = p['enc0.conv2.weight'] / gain
    models = {}
    for dt in (None, 'fp32'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(num_classes=C, dim=3, act_dtype=dt, pretrained=False)
        m.load_named(p)
        models[dt] = m.cuda().eval()
    os.makedirs('model')
    models[None].save_checkpoint(os.path.join('model', 'model.ckpt'))
    vol = _volume(V, 47)
    os.makedirs(os.path.join('data', 'image_volumes'))
    os.makedirs(os.path.join('data', 'predicted_volumes'))
    multiscale.create_multiscale_zarr(vol, os.path.join('data', 'image_volumes', 'a.zarr'), chunk_size=16, shard_size=32)
    want = predict.predict_volume_array(models['fp32'], vol, input_size=S, num_classes=C).cpu().numpy().astype(int)
    predict.predict_volumes(input_size=S, num_classes=C, chunk_size=16, shard_size=32)
    out = capsys.readouterr().out
    assert 'predicting again in fp16x2_wide' in out and 'WARNING' not in out
    got = zarr3.open(os.path.join('data', 'predicted_volumes', 'a.zarr'))['0'][...].astype(int)
    assert got.shape == want.shape and np.abs(got - want).max() <= 1
    assert (got != want).mean() < 0.02


def test_train_model_files_and_learning(tmp_path, monkeypatch):
    """trainer.train_model with injected loaders: loss goes down, model/model.ckpt and the Lightning-style
    metrics.csv appear, a second call resumes from the checkpoint (trainer.py:30-49)."""
    from interactive_unet import trainer, unet
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    imgs = np.stack([_volume((64, 64), 30 + i) for i in range(4)])
    X = torch.tensor(imgs[:, None].astype(np.float32) / 255).half()
    lab = imgs > 127
    y = torch.tensor(np.stack([~lab, lab], 1).astype(np.float32)).half()
    w = torch.ones_like(y)
    train = [(X[:2], y[:2], w[:2]), (X[2:], y[2:], w[2:])]
    val = [(X[:2], y[:2], w[:2])]
    trainer.train_model(1e-3, 2, 6, 1, 2, 'MCC + CE', 'U-Net', 'mit_b0', False, train_loader=train, val_loader=val)
    assert os.path.isfile('model/model.ckpt')
    files = glob.glob('model/history/*/version_0/metrics.csv')
    assert len(files) == 1
    rows = list(csv.DictReader(open(files[0])))
    tr = [float(r['train/Loss']) for r in rows if r['train/Loss']]
    va = [float(r['val/Loss']) for r in rows if r['val/Loss']]
    assert len(tr) == 6 and len(va) == 6 and set(rows[0]) >= {'epoch', 'step', 'train/Dice', 'val/MCC'}
    assert tr[-1] < tr[0] - 0.05, tr
    m = unet.UNet.load_from_checkpoint(checkpoint_path='model/model.ckpt')
    assert m.num_classes == 2 and abs(m.lr - 1e-3) < 1e-12
    trainer.train_model(5e-4, 2, 1, 1, 2, 'Dice + CE', 'U-Net', 'mit_b0', False, train_loader=train, val_loader=val)
    assert len(glob.glob('model/history/*/version_0/metrics.csv')) >= 1 and os.path.isfile('model/model.ckpt')


def test_device_slicer_bit_exact(golden_dir):
    """Slicer.get_slice on a resident uint8 volume (iunet_slice_gather) against (a) the slices the REFERENCE produced
    for the golden cases (tests/golden/slicer.npz, orders 0 and 1, all three planes) and (b) the oracle's scipy path
    on larger random poses, axis-aligned ones and poses that leave the volume: identical bytes."""
    from interactive_unet.slicer import Slicer
    from oracle import slicer_ref
    g = np.load(os.path.join(golden_dir, 'slicer.npz'))
    vol = g['ramp']                  # the volume the reference sliced (make_golden.py: make_slicer)
    vd = torch.tensor(vol).cuda()
    for i in range(int(g['n'])):
        s = Slicer(volume_shape=list(vol.shape))
        s.update_orientation_vectors(g[f's{i}_rv'])
        s.origin = g[f's{i}_origin'].astype(float)
        for axis in range(3):
            for order in (0, 1):
                got = s.get_slice(vd, axis=axis, slice_width=24, order=order).cpu().numpy()
                assert np.array_equal(got, g[f's{i}_slice_a{axis}_o{order}']), (i, axis, order)
    rng = np.random.default_rng(7)
    V = (70, 96, 83)
    big = rng.integers(0, 256, V, dtype=np.uint8)
    bd = torch.tensor(big).cuda()
    np.random.seed(11)
    for trial in range(12):
        s = Slicer(volume_shape=list(V))
        if trial % 4 == 3:
            s.randomize(sampling_mode='grid')                       # axis-aligned, integer normal
        else:
            s.randomize(sampling_mode='random', origin_shift_range=1.0 if trial % 2 else 0.8)
        if trial == 5:
            s.origin = np.array([-40.0, 10.0, 70.0])                # mostly outside the volume
        for axis in range(3):
            for order in (0, 1):
                for sw in (64, 129):
                    want = slicer_ref.get_slice(big, s.u, s.v, s.w, s.origin, axis=axis, slice_width=sw, order=order,
                                                sampling_axis=s.sampling_axis)
                    got = s.get_slice(bd, axis=axis, slice_width=sw, order=order).cpu().numpy()
                    assert np.array_equal(got, want), (trial, axis, order, sw, np.abs(got.astype(int) - want).max())


def test_data_parallel_step_over_rccl_single_rank():
    """The N > 1 training path of bench.py (flat-gradient all-reduce through torch.distributed 'nccl' = RCCL, gradient
    divided by the world size) on a one-rank process group: same loss sequence as the plain engine."""
    import torch.distributed as dist
    from interactive_unet.train_engine import TrainEngine
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    import socket
    with socket.socket() as sk:                      # a free port for the one-rank rendezvous
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        g = torch.Generator().manual_seed(41)
        X = torch.randint(1, 255, (1, 1, 16, 32, 32), dtype=torch.uint8, generator=g)
        lab = X > 127
        y = torch.cat([~lab, lab], 1).half()
        losses = []
        for pg in (None, dist.group.WORLD):
            model, _ = _model(3, 2, seed=7, dtype='bf16')
            te = TrainEngine(model.train(), lr=1e-3, loss_kind='mcc_ce', process_group=pg)
            losses.append([te.train_step(X, y, None)['Loss'] for _ in range(4)])
            # from the second step both run the C-sequenced step; the data-parallel one through iunet_train_forward_backward_hooks, whose
            # host callback starts the gradient buckets between the backward's launches
            assert getattr(te, '_h', None) is not None
        assert np.allclose(losses[0], losses[1], rtol=0, atol=1e-6), losses
        assert losses[0][2] < losses[0][0]
    finally:
        dist.destroy_process_group()


def test_multiscale_pyramid_bit_exact(golden_dir):
    """utils.resize_volume / multiscale_levels on the device (iunet_zoom_nearest_u8) against (a) the volumes the REFERENCE's
    resize_volume produced (tests/golden/multiscale.npz: several blocks per axis, ragged blocks, scipy's constant fill,
    4-D prediction volumes), device-resident and host-staged, and (b) the oracle on larger volumes and a whole pyramid:
    identical bytes.  Odd blocks whose zoomed shape does not fit their slot raise like numpy does in the reference."""
    from interactive_unet import utils
    from oracle import multiscale_ref as mr
    g = np.load(os.path.join(golden_dir, 'multiscale.npz'))
    for i in range(int(g['n'])):
        src, want, block = g[f'c{i}_src'], g[f'c{i}_dst'], int(g[f'c{i}_block'])
        dst = torch.full(want.shape, 7, dtype=torch.uint8, device='cuda')
        utils.resize_volume(torch.tensor(src).cuda(), dst, scale=0.5, block_size=block, order=0)
        assert np.array_equal(dst.cpu().numpy(), want), i
        host = np.full_like(want, 7)                                    # host (Zarr-like) arrays: blocks staged through the GPU
        utils.resize_volume(src, host, scale=0.5, block_size=block, order=0)
        assert np.array_equal(host, want), i
    rng = np.random.default_rng(11)
    for shape, block in [((130, 96, 258), 64), ((96, 64, 80, 2), 32), ((64, 64, 64, 3), 64), ((33, 35, 37), 512)]:
        src = rng.integers(1, 256, shape, dtype=np.uint8)
        want = np.full(tuple(int(x * 0.5) for x in shape), 9, dtype=np.uint8)
        try:
            mr.resize_volume(src, want, 0.5, block)
        except ValueError:
            with pytest.raises(ValueError):
                utils.resize_volume(torch.tensor(src).cuda(), torch.empty(want.shape, dtype=torch.uint8, device='cuda'), 0.5, block)
            continue
        dst = torch.full(want.shape, 9, dtype=torch.uint8, device='cuda')
        utils.resize_volume(torch.tensor(src).cuda(), dst, scale=0.5, block_size=block)
        assert np.array_equal(dst.cpu().numpy(), want), shape
    for scale, shape, block in [(0.25, (128, 64, 192), 64), (0.25, (96, 96, 96, 4), 32), (0.75, (64, 32, 48), 16)]:      # other zoom factors
        src = rng.integers(1, 256, shape, dtype=np.uint8)
        want = np.full(tuple(int(x * scale) for x in shape), 5, dtype=np.uint8)
        mr.resize_volume(src, want, scale, block)
        dst = torch.full(want.shape, 5, dtype=torch.uint8, device='cuda')
        utils.resize_volume(torch.tensor(src).cuda(), dst, scale=scale, block_size=block)
        assert np.array_equal(dst.cpu().numpy(), want), (scale, shape)
    # a whole pyramid (add_multiscales on arrays): 256 x 192 x 320 with 32^3 chunks and 64^3 shards -> 3 levels
    vol = rng.integers(0, 256, (256, 192, 320), dtype=np.uint8)
    want = mr.multiscale_levels(vol, (32,) * 3, (64,) * 3)
    got = utils.multiscale_levels(torch.tensor(vol).cuda(), (32,) * 3, (64,) * 3)
    assert len(got) == len(want) == 3
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)
    with pytest.raises(ValueError):
        utils.resize_volume(torch.tensor(vol).cuda(), got[0], order=1)


def test_device_update_volume_bit_exact(golden_dir):
    """Slicer.update_volume on a resident uint8 volume (iunet_slice_scatter) against (a) the volumes the REFERENCE wrote for
    the golden poses (tests/golden/slicer.npz) and (b) the numpy path of the host mirror (pinned by the same goldens) on poses
    with many duplicate targets: slices larger than the volume (everything outside is clipped onto the faces), oblique
    planes, 4-D [Z, Y, X, C] volumes.  Identical bytes: the last pixel in row-major order wins, as in numpy."""
    from interactive_unet.slicer import Slicer
    g = np.load(os.path.join(golden_dir, 'slicer.npz'))
    for i in range(int(g['n'])):
        s = Slicer(volume_shape=[32, 32, 32])
        s.update_orientation_vectors(g[f's{i}_rv'])
        s.origin = g[f's{i}_origin'].copy()
        vol = torch.tensor(g['vol']).cuda()
        out = s.update_volume(g[f's{i}_upd_data'], vol, axis=1)
        assert out is vol and np.array_equal(vol.cpu().numpy(), g[f's{i}_upd_vol']), i
    rng = np.random.default_rng(21)
    for trial in range(12):
        shape = [int(v) for v in rng.integers(20, 70, 3)]
        C = [None, 3][trial % 2]
        s = Slicer(volume_shape=shape)
        s.update_orientation_vectors(rng.normal(size=3))
        s.origin = np.array(shape) * rng.random(3)
        sw = int(rng.choice([16, 31, 64, 100, 129]))
        data = rng.integers(0, 256, (sw, sw) if C is None else (sw, sw, C), dtype=np.uint8)
        base = rng.integers(0, 256, shape if C is None else shape + [C], dtype=np.uint8)
        for axis in (0, 1, 2):
            want = s.update_volume(data, base.copy(), axis=axis)                 # numpy path (host mirror)
            got = s.update_volume(data, torch.tensor(base).cuda(), axis=axis)
            assert np.array_equal(got.cpu().numpy(), want), (trial, axis)


def _annotation(rng, H, W, C, ch=None):
    image = rng.integers(0, 256, (H, W) if ch is None else (H, W, ch), dtype=np.uint8)
    image[rng.random(image.shape) < 0.15] = 0
    mask = (np.eye(C, dtype=np.uint8)[rng.integers(0, C, (H, W))] * 255).astype(np.uint8)
    weight = rng.integers(0, 256, (H, W), dtype=np.uint8)
    return image, mask, weight


def test_device_batch_producer_bit_exact(golden_dir):
    """loader.UNetDataset.batch (iunet_augment_batch) against the oracle's loader restatement for given transform parameters:
    (a) without augmentation the reference's own normalised annotations (tests/golden/loader.npz) in float16; (b) with
    augmentation: random angles, the rot90 fast paths, flips, square / non-square / multi-channel annotations, mixed sizes
    in one batch -- identical float16 bits for image, mask and weight."""
    from interactive_unet import loader
    from oracle import loader_ref as lr
    g = np.load(os.path.join(golden_dir, 'loader.npz'))
    for k in range(int(g['n'])):
        ann = loader.annotations_from_arrays([(g[f'c{k}_image'], g[f'c{k}_mask'], g[f'c{k}_weight'])])
        X, y, w = loader.UNetDataset(ann, None, augment=False)[0]
        for got, name in zip((X, y, w), ('image', 'mask', 'weight')):
            want = torch.from_numpy(g[f'c{k}_{name}_f']).to(torch.float16)
            assert got.dtype == torch.float16 and torch.equal(got.cpu(), want), (k, name)
    rng = np.random.default_rng(4)
    fixed = [0.0, 180.0, 90.0, -90.0, 360.0, 270.0]
    for trial in range(10):
        C, ch = [(2, None), (3, None), (2, 3)][trial % 3]
        sizes = [(512, 512), (300, 400), (256, 256), (611, 389)]
        samples = [_annotation(rng, *sizes[(trial + b) % 4], C, ch) for b in range(3)]
        ds = loader.UNetDataset(loader.annotations_from_arrays(samples), None, augment=True)
        params = []
        for b, (image, _, _) in enumerate(samples):
            H, W = image.shape[:2]
            ang = fixed[trial] if (trial < len(fixed) and b == 0) else float(rng.uniform(-360, 360))
            crop = lr.resized_crop_params(H, W, lambda a, c: float(rng.uniform(a, c)), lambda n: int(rng.integers(n)))
            params.append((bool(rng.integers(2)), bool(rng.integers(2)), ang, crop))
        X, y, w = ds.batch([0, 1, 2], params=params)
        assert X.shape == (3, 1 if ch is None else ch, 512, 512) and y.shape == w.shape == (3, C, 512, 512)
        for b, (image, mask, weight) in enumerate(samples):
            want = lr.get_item(*lr.normalise(image, mask, weight), *params[b])
            for got, ref, name in zip((X[b], y[b], w[b]), want, ('image', 'mask', 'weight')):
                assert torch.equal(got.cpu(), ref), (trial, b, name)
    # the loader: epoch length, batch shapes, determinism under a seeded generator, shuffling
    samples = [_annotation(rng, 256, 256, 2) for _ in range(5)]
    ann = loader.annotations_from_arrays(samples)
    a = [t[0].clone() for t in loader.get_data_loader(batch_size=2, annotations=ann, generator=torch.Generator().manual_seed(5))]
    b = [t[0].clone() for t in loader.get_data_loader(batch_size=2, annotations=ann, generator=torch.Generator().manual_seed(5))]
    assert len(a) == 3 and a[0].shape == (2, 1, 512, 512) and a[2].shape[0] == 1
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    val = list(loader.get_data_loader(set_type='val', batch_size=5, augment=False, shuffle=False, annotations=ann))
    assert len(val) == 1 and val[0][0].shape == (5, 1, 256, 256)
    with pytest.raises(RuntimeError):
        loader.UNetDataset(loader.annotations_from_arrays([_annotation(rng, 64, 64, 2), _annotation(rng, 64, 80, 2)]), None).batch([0, 1])


def test_train_model_reads_annotation_files(tmp_path, monkeypatch):
    """trainer.train_model exactly as app.py:719 calls it (positional arguments, no loaders injected): the annotations are
    read from data/{train,val}/{images,masks,weights} (colour masks decoded like utils.colored_to_categorical), batches come
    from the device producer (augmented 512 x 512 for training, as-is for validation), checkpoint and history appear."""
    from PIL import Image
    from interactive_unet import trainer, loader
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(8)
    for split, n in (('train', 3), ('val', 2)):
        for sub in ('images', 'masks', 'weights'):
            os.makedirs(os.path.join('data', split, sub))
        for k in range(n):
            img = _volume((96, 96), 50 + k)
            img[:4] = 0                                                    # a black border: mask / weight are zeroed there
            cls = (img > 127).astype(int) + 1                             # palette colours 1 and 2; colour 0 = unlabelled
            cls[rng.random(cls.shape) < 0.1] = 0
            Image.fromarray(img).save(os.path.join('data', split, 'images', f'{k:04d}.tiff'))
            Image.fromarray(loader.COLORS[cls]).save(os.path.join('data', split, 'masks', f'{k:04d}.tiff'))
            Image.fromarray(np.full((96, 96), 255, np.uint8)).save(os.path.join('data', split, 'weights', f'{k:04d}.tiff'))
    ann = loader.load_annotations('train')
    assert len(ann) == 3 and ann[0][0].shape == (96, 96, 1) and ann[0][1].shape == (96, 96, 2) and ann[0][0].is_cuda
    mask0, weight0 = loader.colored_to_categorical(np.asarray(Image.open('data/train/masks/0000.tiff')))
    assert set(np.unique(mask0)) <= {0, 255} and np.array_equal(weight0 == 255, mask0.sum(-1) == 255)
    trainer.train_model(1e-3, 2, 2, 1, 2, 'MCC + CE', 'U-Net', 'mit_b0', False)
    assert os.path.isfile('model/model.ckpt')
    rows = list(csv.DictReader(open(glob.glob('model/history/*/version_0/metrics.csv')[0])))
    assert len([r for r in rows if r['train/Loss']]) == 2 and len([r for r in rows if r['val/Loss']]) == 2


def test_suggestor_contract_and_learning():
    """suggestor.make_suggestions (suggestor.py:43-116) on the native train step: return types and shapes, the single-class
    shortcut (pure numpy in the reference: identical bytes), model reuse while the class count stays, a new model when it
    changes, and that the fine-tune learns an easy two-region slice from sparse scribbles (the network differs from the
    reference's smp mobilenet -- parity unpinned there -- so the learning check is a property, not a comparison)."""
    from interactive_unet import suggestor, loader
    rng = np.random.default_rng(12)
    S = 128
    truth = np.zeros((S, S), int)
    truth[:, S // 2:] = 1
    img = np.clip(np.where(truth == 1, 200, 60) + rng.normal(0, 12, (S, S)), 1, 255).astype(np.uint8)
    feats = (img / 255).astype('float32')[None, None, :, :]                        # app.py:311
    mask = np.zeros((S, S, 3), np.uint8)                                            # black = unlabelled
    for r in (20, 64, 100):                                                          # sparse scribbles in both regions
        mask[r:r + 3, 8:48] = loader.COLORS[1]
        mask[r:r + 3, 80:120] = loader.COLORS[2]
    one = np.zeros((S, S, 3), np.uint8)
    one[10:20, 10:20] = loader.COLORS[3]
    sug, m0 = suggestor.make_suggestions(feats, one, model='kept')
    assert m0 == 'kept' and sug.shape == (S, S, 3) and sug.dtype == np.uint8 and (sug == loader.COLORS[3]).all()
    gen = torch.Generator().manual_seed(3)
    sug, model = suggestor.make_suggestions(feats, mask, lr=1e-3, steps=60, generator=gen)
    assert sug.shape == (S, S, 3) and sug.dtype == np.uint8 and isinstance(model, suggestor.Suggestor)
    colours = {tuple(c) for c in sug.reshape(-1, 3)}
    assert colours <= {tuple(loader.COLORS[1]), tuple(loader.COLORS[2])}
    pred = (sug == loader.COLORS[2]).all(-1).astype(int)
    assert (pred == truth).mean() > 0.9, (pred == truth).mean()
    sug2, model2 = suggestor.make_suggestions(feats, mask, model=model, generator=gen)           # defaults: lr 1e-4, 30 steps
    assert model2 is model and ((sug2 == loader.COLORS[2]).all(-1).astype(int) == truth).mean() > 0.8
    mask3 = mask.copy()
    mask3[60:70, 56:72] = loader.COLORS[4]
    sug3, model3 = suggestor.make_suggestions(feats, mask3, steps=2, model=model, generator=gen)
    assert model3 is not model and model3.num_classes == 3 and sug3.shape == (S, S, 3)


def test_suggestor_trajectory_against_the_oracle_train_loop():
    """SURVEY 8f rank 4 (suggestor.py:78-103): the per-brush-stroke fine-tune is 30 steps of {augment, forward, MCC + CE with the
    annotation mask as weight, backward, AdamW}.  The network behind it is unpinned (smp mobilenet absent), so what CAN be pinned
    is the loop: the native run records the augmented batch and the loss of every step; the oracle (oracle/unet_ref.py forward in
    training mode with the HIP path's fp16 rounding points + the host metrics' mcc_ce_loss + torch autograd + the restated AdamW)
    replays the same batches from the same initial weights.  The two loss trajectories must track each other step by step."""
    from interactive_unet import suggestor, loader, metrics as host_metrics
    rng = np.random.default_rng(21)
    S, steps, lr = 64, 30, 1e-4
    truth = np.zeros((S, S), int)
    truth[S // 2:, :] = 1
    img = np.clip(np.where(truth == 1, 190, 70) + rng.normal(0, 15, (S, S)), 1, 255).astype(np.uint8)
    feats = (img / 255).astype('float32')[None, None]
    mask = np.zeros((S, S, 3), np.uint8)
    mask[8:12, 6:40] = loader.COLORS[1]
    mask[50:54, 20:60] = loader.COLORS[2]
    model = suggestor.Suggestor(1, 2)
    model.reset_parameters(seed=4)
    p0 = {k: v.detach().clone() for k, v in model.named_tensors().items()}
    trace = []
    sug, model = suggestor.make_suggestions(feats, mask, lr=lr, steps=steps, model=model.cuda(), generator=torch.Generator().manual_seed(9),
                                            trace=trace)
    assert len(trace) == steps and sug.shape == (S, S, 3)
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p0.items()}
    m = {k: torch.zeros_like(v) for k, v in pr.items()}
    v = {k: torch.zeros_like(t) for k, t in pr.items()}
    names = [k for k, t in pr.items() if t.requires_grad]
    worst = 0.0
    for it, (x, y, w, loss_native) in enumerate(trace):
        probs = unet_ref.forward(pr, x, dim=2, training=True, act_dtype=torch.float16)
        loss = host_metrics.mcc_ce_loss(probs, y, w, axes=[0, 2, 3])
        grads = torch.autograd.grad(loss, [pr[k] for k in names])
        with torch.no_grad():
            unet_ref.adamw_step({k: t.data for k, t in pr.items()}, dict(zip(names, grads)), m, v, it + 1, lr)
        d = abs(loss.item() - loss_native)
        worst = max(worst, d)
        assert d <= (5e-3 if it == 0 else 2e-2), (it, loss.item(), loss_native)
    first, last = trace[0][3], trace[-1][3]
    print(f'suggestor trajectory: native loss {first:.4f} -> {last:.4f} over {steps} steps; max |native - oracle| per step = {worst:.2e}')
    # after the replay the two runs have moved the weights the same way (AdamW normalises the gradient, so an entry whose tiny
    # gradient differs in the fp16 noise can move the other way: compare directions, not entries)
    nat = {k: t.detach().cpu() for k, t in model.named_tensors().items()}
    for k in ('enc0.conv1.weight', 'dec0.conv2.weight', 'head.weight'):
        a, b = (nat[k] - p0[k]).reshape(-1).double(), (pr[k].detach() - p0[k]).reshape(-1).double()
        cos = float(a @ b / (a.norm() * b.norm()))
        assert cos >= 0.7, (k, cos)
