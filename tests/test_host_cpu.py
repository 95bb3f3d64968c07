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
