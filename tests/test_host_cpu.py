"""Host-side logic of the drop-in package against the reference goldens (CPU only): block
grid, windows, shard grid, padded blocks, Slicer geometry, loss selectors, checkpoints."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def G(golden_dir):
    return {k: np.load(os.path.join(golden_dir, k + '.npz')) for k in ('losses', 'predict', 'slicer')}


def test_predict_helpers_match_reference(G):
    from interactive_unet import predict
    g = G['predict']
    for k in range(int(g['n_bc'])):
        a = g[f'bc{k}_args']
        b, pb, lb = predict.get_block_coordinates(a[:3], int(a[3]), a[4] / 100.0)
        assert np.array_equal(b, g[f'bc{k}_b']) and np.array_equal(pb, g[f'bc{k}_pb']) and np.array_equal(lb, g[f'bc{k}_lb'])
    for i, c in enumerate(g['pad_coords']):
        assert np.array_equal(predict.get_padded_block(g['pad_vol'], *c), g[f'pad{i}'])
    for S in (8, 16, 32):
        assert np.array_equal(predict.gaussian_3d(S), g[f'gauss{S}'])
        assert np.array_equal(predict.hanning_3d(S), g[f'hann{S}'])
    assert np.array_equal(predict.get_shard_coordinates([300, 260, 129], 128), g['shards_300_260_129_128'])
    for n in (1, 2, 7, 16):
        assert np.array_equal(predict.reflect_index(g['reflect_idx'], n), g[f'reflect_{n}'])


def test_slicer_class_matches_reference(G):
    from interactive_unet.slicer import Slicer
    g = G['slicer']
    for i in range(int(g['n'])):
        s = Slicer(volume_shape=[32, 32, 32])
        s.update_orientation_vectors(g[f's{i}_rv'])
        s.origin = g[f's{i}_origin'].copy()
        for name, val in (('rotvec', s.rot_vec), ('rotmat', s.rot_mat), ('u', s.u), ('v', s.v), ('w', s.w)):
            assert np.array_equal(val, g[f's{i}_{name}'])
        assert np.array_equal(s.get_interpolation_coords(8), g[f's{i}_coords8'])
        for axis in (0, 1, 2):
            for order in (0, 1):
                assert np.array_equal(s.get_slice(g['ramp'], axis=axis, slice_width=24, order=order),
                                      g[f's{i}_slice_a{axis}_o{order}'])
        assert np.array_equal(s.update_volume(g[f's{i}_upd_data'], g['vol'].copy(), axis=1), g[f's{i}_upd_vol'])
        s2 = Slicer(volume_shape=[8, 8, 8])
        s2.from_dict(s.to_dict())
        assert np.array_equal(s2.u, g[f's{i}_rt_u'])


def test_metric_functions_match_reference(G):
    from interactive_unet import metrics
    g = G['losses']
    fns = {'ce': metrics.crossentropy_loss, 'dice': metrics.dice_loss, 'iou': metrics.iou_loss, 'mcc': metrics.mcc_loss,
           'dice_ce': metrics.dice_ce_loss, 'iou_ce': metrics.iou_ce_loss, 'mcc_ce': metrics.mcc_ce_loss}
    for c in range(int(g['n_cases'])):
        key = f'c{c}'
        p, y = torch.tensor(g[key + '_p']), torch.tensor(g[key + '_y'])
        w = torch.tensor(g[key + '_w']) if key + '_w' in g else None
        axes = list(g[key + '_axes'])
        for kind, fn in fns.items():
            assert fn.native_kind == kind
            assert abs(fn(p, y, w, axes=axes).item() - float(g[f'{key}_{kind}'])) < 1e-10
    assert metrics.loss_name_to_function('MCC + CE') is metrics.mcc_ce_loss


def test_unet_signature_and_checkpoint_roundtrip(tmp_path):
    import inspect
    import warnings
    from interactive_unet import unet, metrics
    sig = inspect.signature(unet.UNet.__init__)
    names = list(sig.parameters)[1:8]
    assert names == ['lr', 'num_channels', 'num_classes', 'loss_function', 'architecture', 'encoder_name', 'pretrained']
    assert sig.parameters['lr'].default == 0.0001 and sig.parameters['architecture'].default == 'U-Net'
    assert sig.parameters['loss_function'].default is metrics.mcc_ce_loss
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = unet.UNet(lr=3e-4, num_classes=3, loss_function=metrics.dice_ce_loss)
    assert sum(p.numel() for p in m.parameters()) == 1926466 + 33      # 1.926 M for 2 classes (1.925 M) + one more head row
    with pytest.raises(NotImplementedError):
        unet.UNet(architecture='PSPNet', pretrained=False)
    with pytest.raises(RuntimeError):
        m.engine('eval')                                   # CPU module: no fallback
    path = str(tmp_path / 'model.ckpt')
    m.save_checkpoint(path)
    m2 = unet.UNet.load_from_checkpoint(checkpoint_path=path)
    assert m2.lr == 3e-4 and m2.num_classes == 3 and m2.loss_function is metrics.dice_ce_loss
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_train_model_signature_matches_reference():
    import inspect
    from interactive_unet import trainer
    names = list(inspect.signature(trainer.train_model).parameters)[:11]
    assert names == ['lr', 'batch_size', 'epochs', 'num_channels', 'num_classes', 'loss_function_name', 'architecture',
                     'encoder_name', 'pretrained', 'reslice', 'reslice_factor']       # app.py:697-719 passes them positionally


def test_engine_auto_host_logic(monkeypatch):
    """engine_auto.EngineAuto's host side without a GPU: policy from the argument / IUNET_X2M, the calibration tile, the re-calibration
    schedule over weight loads, the adoption of a figure at the next load, and the range steps."""
    import torch
    from interactive_unet.engine_auto import EngineAuto, THRESHOLD
    monkeypatch.delenv('IUNET_X2M', raising=False)
    assert EngineAuto(dim=2, device='cpu').policy == 'auto'
    monkeypatch.setenv('IUNET_X2M', '0')
    assert EngineAuto(dim=2, device='cpu').policy == 'fp16x2' and EngineAuto(dim=2, device='cpu', policy='x2m').policy == 'x2m'
    monkeypatch.setenv('IUNET_X2M', '1')
    assert EngineAuto(dim=3, device='cpu').mode == 'x2m'
    monkeypatch.delenv('IUNET_X2M')
    with pytest.raises(ValueError):
        EngineAuto(dim=2, device='cpu', policy='fp8')
    e = EngineAuto(dim=3, levels=4, device='cpu', recal_every=4)
    assert e._crop(128, 128, 128) == (64, 64, 64) and e._crop(16, 32, 200) == (16, 32, 64) and e._crop(8, 8, 8) == (8, 8, 8)
    e5 = EngineAuto(dim=2, levels=5, device='cpu')
    assert e5._crop(1, 512, 48) == (1, 256, 48) and e5._crop(1, 16, 16) == (1, 16, 16)

    class _Form:
        loads = 0

        def load_eval(self, params):
            _Form.loads += 1
    e._form = lambda name: _Form()
    figures = iter([1e-4, 9e-4, 2e-4])
    e._measure = lambda x, xs, D, H, W: (torch.tensor([next(figures), 3.0]), (64, 64, 64))
    with pytest.raises(RuntimeError):
        e.calibrate(object(), None, 64, 64, 64)
    trace = []
    for load in range(1, 11):
        e.load_eval({})
        if e._due():
            e.calibrate(object(), None, 64, 64, 64, blocking=e.mode is None)
        trace.append((load, e.form, e.calibrations))
    # load 1: blocking -> x2m at once; load 5: measured 9e-4, adopted at load 6 -> fp16x2; load 9: 2e-4, adopted at load 10 -> x2m
    assert [f for _, f, _ in trace] == ['x2m'] * 5 + ['fp16x2'] * 4 + ['x2m']
    assert [c for _, _, c in trace] == [1, 1, 1, 1, 2, 2, 2, 2, 3, 3]
    assert e.calibration['diff'] == pytest.approx(2e-4) and e.calibration['threshold'] == THRESHOLD and e.describe()['form'] == 'x2m'
    assert e.widen() and e.form == 'fp16x2_wide' and not e._due()
    assert e.widen() and e.form == 'fp32' and not e.widen()


def test_block_grid_and_padded_blocks_against_the_oracle_on_random_shapes():
    """Beyond the reference-generated goldens: the native block grid (predict.py:362-411, truncating float -> int), reflect-padded blocks
    (predict.py:291-316) and the sharded partition on random volume shapes / block sizes / overlaps against the oracle restatement (itself
    pinned by the goldens), plus the invariants a prediction relies on: every voxel covered, local = clipped - padded, blocks inside the
    padded extent."""
    from hypothesis import given, settings, strategies as st
    from interactive_unet import predict, shard
    from oracle import predict_ref

    @settings(max_examples=60, deadline=None)
    @given(st.tuples(st.integers(9, 140), st.integers(9, 140), st.integers(9, 140)), st.sampled_from([8, 16, 24, 32, 64]),
           st.sampled_from([0.0, 0.125, 0.25, 0.5]), st.integers(1, 8))
    def check(V, S, overlap, world):
        b, pb, lb = predict.get_block_coordinates(np.array(V), S, overlap)
        wb, wpb, wlb = predict_ref.get_block_coordinates(V, S, overlap)
        assert np.array_equal(b, wb) and np.array_equal(pb, wpb) and np.array_equal(lb, wlb)
        if len(pb) == 0:          # a volume thinner than overlap * S along an axis: the reference's grid is empty there (predict.py:376), and so is ours
            assert min(V) <= overlap * S
            return
        assert (pb[:, 3:] - pb[:, :3] == S).all() and np.array_equal(lb[:, :3], b[:, :3] - pb[:, :3]) and np.array_equal(lb[:, 3:], b[:, 3:] - pb[:, :3])
        cover = np.zeros(V, dtype=np.int32)
        for c in b:
            cover[c[0]:c[3], c[1]:c[4], c[2]:c[5]] += 1
        assert cover.min() >= 1
        runs = shard.partition_blocks(len(pb), world)
        assert runs[0][0] == 0 and runs[-1][1] == len(pb) and all(a[1] == c[0] for a, c in zip(runs, runs[1:]))
        rng = np.random.default_rng(sum(V) + S)
        vol = rng.integers(0, 256, size=V, dtype=np.uint8)
        for k in rng.integers(0, len(pb), size=min(3, len(pb))):
            if (np.array(V) >= 2).all() and (b[k][3:] - b[k][:3] >= 2).all():          # np.pad(reflect) needs two planes to mirror
                assert np.array_equal(predict.get_padded_block(vol, *pb[k]), predict_ref.get_padded_block(vol, *[int(v) for v in pb[k]]))
    check()
