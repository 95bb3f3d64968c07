"""The default prediction mode on TRAINED networks.

x2m's logit error is a fraction of the logit scale, and a converged segmentation network is confident: these tests train the canonical
2-D and 3-D nets natively on separable synthetic labels until max |logit| >= 20, then hold whatever form `UNet()` selects
(engine_auto.EngineAuto: x2m or fp16x2 by calibration) against the CPU fp32 oracle at the headline sizes -- 512^2 slices, one 128^3
chunk -- with the gate of tests/test_gpu_parity.py: logits <= 1e-3 absolute, class map equal outside the tie band (population printed).
The selection rule, its fallback, the automatic re-run after a saturated forward and the asynchronous re-calibration inside a training
loop are exercised one by one.
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref
from tests.test_gpu_parity import TOL, _assert_fp32_mode, _compare, _forward, _labels, _smooth


def _unet(**kw):
    from interactive_unet.unet import UNet
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return UNet(pretrained=False, **kw)


def _train_until_confident(dim, target=20.0, max_steps=6000, check_every=100, lr=8e-3, seed=0, cooldown=150):
    """Native fp16 training (MCC+CE, the reference's default loss) on label = (smooth image > 127) until the 16-bit forward's max |logit|
    on a held-out tile reaches `target` AND the eval-mode network segments its training tiles (Dice >= 0.97): bursts at `lr` until the
    logits are there, then `cooldown` steps at lr / 20 (the BatchNorm running statistics catch up with the weights), then the check.
    -> ({name: cpu tensor}, steps, logit scale)."""
    from interactive_unet.train_engine import TrainEngine
    from interactive_unet.engine import Engine
    dev = torch.device('cuda')
    tshape, B = ((256, 256), 8) if dim == 2 else ((64, 64, 64), 8)       # (3-D: 8 chunks per step -- with 2 the bottom level's batch statistics are 1 024 values per channel and the eval-mode network lags the training-mode one)
    m = _unet(lr=lr, dim=dim, act_dtype='fp16')
    m.reset_parameters(seed=seed)
    m = m.to(dev)
    te = TrainEngine(m, lr=lr, loss_kind='mcc_ce', weight_decay=0.0)
    imgs = np.stack([_smooth(tshape, 100 + i, sigma=6) for i in range(4 * B)])[:, None]
    X = torch.tensor(imgs).to(dev)
    lab = X > 127
    Y = torch.cat([~lab, lab], 1).to(torch.float16)
    Wt = torch.ones_like(Y)
    ev = torch.tensor(_smooth(tshape, 999, sigma=6))[None, None].to(dev)
    D, H, W = tshape if dim == 3 else (1,) + tshape
    vox = D * H * W
    lg = torch.empty((1, 2) + tshape, device=dev)
    e16 = Engine(dim=dim, act_dtype=torch.float16)
    scale, best, row, s = 0.0, None, None, 0

    def steps(n):
        nonlocal s
        for _ in range(n):
            s += 1
            i = (s % 4) * B
            te.train_step(X[i:i + B], Y[i:i + B], Wt[i:i + B], sync=False)

    def logit_scale():
        e16.load_eval(m.named_tensors())
        e16.infer(ev, (vox, vox, H * W, W, 1), 1, D, H, W, logits=lg)
        v = float(lg.abs().max())
        assert np.isfinite(v), 'training diverged'
        return v
    while s < max_steps:
        te.lr = lr
        steps(check_every)
        scale = logit_scale()
        if scale < 1.1 * target:
            continue
        te.lr = lr / 20
        steps(cooldown)
        scale = logit_scale()
        row = te.eval_step(X[:B], Y[:B], Wt[:B])                  # eval-mode BatchNorm: the network the prediction runs
        print(f'    [{dim}-D training] step {s}: max |logit| {scale:.1f}, eval-mode Dice {row["Dice"]:.4f}')
        if scale >= target and row['Dice'] >= 0.97:               # confident AND right: a segmentation network, not a diverged one
            best = s
            break
    assert best is not None, f'after {s} steps: max |logit| {scale:.1f}, last eval {row}'
    print(f'[trained {dim}-D] {best} steps, max |logit| {scale:.1f} on the held-out tile; training-set Dice {row["Dice"]:.4f} MCC {row["MCC"]:.4f}')
    return {k: t.detach().float().cpu().clone() for k, t in m.named_tensors().items()}, best, scale


@pytest.fixture(scope='module')
def trained2d():
    return _train_until_confident(2)


@pytest.fixture(scope='module')
def trained3d():
    return _train_until_confident(3, lr=1e-3, check_every=50)


def _default_mode_against_oracle(dim, p, shape, N, seed):
    """UNet() as the reference constructs it, trained weights loaded -> parity figures of the form it selects, plus both pinned forms."""
    from interactive_unet.engine_auto import EngineAuto
    img = np.stack([_smooth(shape, seed * 100 + i, sigma=6) for i in range(N)])[:, None]
    x = torch.tensor(img)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=dim)
    y_true = _labels(img, 2)
    assert ref.abs().max().item() >= 15.0             # the confident regime the test is about (a different tile than the training check's)
    m = _unet(dim=dim)
    m.load_named(p)
    m = m.cuda().eval()
    eng = m.engine('eval')
    assert isinstance(eng, EngineAuto) and eng.policy == 'auto'
    r = _compare(f'UNet() default, trained {dim}-D {N} x {shape}', *_forward(eng, x.cuda(), dim, 2), ref, y_true)
    d = eng.describe()
    print(f'    selected form {d["form"]}: calibration max |x2m - fp16x2| = {d["calibration_max_abs_logit_diff_x2m_vs_fp16x2"]:.2e} on tile '
          f'{d["calibration_tile"]} (threshold {d["threshold"]:.0e}), logit scale of the tile {d["calibration_logit_scale"]:.1f}')
    assert d['form'] in ('x2m', 'fp16x2') and eng.calibrations == 1
    assert (d['form'] == 'x2m') == (d['calibration_max_abs_logit_diff_x2m_vs_fp16x2'] <= d['threshold'])
    assert not eng.saturated()
    _assert_fp32_mode(r)
    res = {'auto': r, 'form': d['form']}
    for policy in ('x2m', 'fp16x2'):
        e = EngineAuto(dim=dim, policy=policy)
        e.load_eval({k: v.cuda() for k, v in p.items()})
        res[policy] = _compare(f'pinned {policy}, trained {dim}-D', *_forward(e, x.cuda(), dim, 2), ref, y_true)
        assert e.calibrations == 0
        del e
        torch.cuda.empty_cache()
    _assert_fp32_mode(res['fp16x2'])                   # the fallback form holds the gate at this logit scale with a wide margin
    assert res['fp16x2']['err'] <= 2e-4
    # what the calibration measures IS x2m's error: the on-device difference to fp16x2 tracks the distance to the CPU oracle
    assert abs(res['x2m']['err'] - d['calibration_max_abs_logit_diff_x2m_vs_fp16x2']) <= 0.6 * res['x2m']['err'] + 1e-4
    if d['form'] == 'x2m':
        assert res['x2m']['err'] <= TOL
    return res


def test_trained_2d_network_at_512_squared(trained2d):
    p, steps, scale = trained2d
    _default_mode_against_oracle(2, p, (512, 512), 2, seed=21)


def test_trained_3d_network_at_one_128_cubed_chunk(trained3d):
    p, steps, scale = trained3d
    _default_mode_against_oracle(3, p, (128, 128, 128), 1, seed=22)


def test_selection_falls_back_to_fp16x2_where_x2m_misses_the_gate():
    """A head 40 times larger: logit scale ~150, where x2m's 3-7e-5 of the scale is several 1e-3.  The calibration must see that and
    every forward must run in fp16x2, which holds 1e-3 there."""
    from interactive_unet.engine_auto import EngineAuto
    p = unet_ref.init_params(dim=2, ncls=2, seed=6, randomize_bn=True)
    p['head.weight'] = p['head.weight'] * 40.0
    shape = (256, 256)
    img = np.stack([_smooth(shape, 700 + i, sigma=6) for i in range(2)])[:, None]
    x = torch.tensor(img)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=2)
    y_true = _labels(img, 2)
    pc = {k: v.cuda() for k, v in p.items()}
    pinned = EngineAuto(dim=2, policy='x2m')
    pinned.load_eval(pc)
    rx = _compare('pinned x2m, head x 40', *_forward(pinned, x.cuda(), 2, 2), ref, y_true)
    assert rx['err'] > TOL, 'the case is meant to be one x2m misses'
    eng = EngineAuto(dim=2)
    eng.load_eval(pc)
    r = _compare('auto, head x 40', *_forward(eng, x.cuda(), 2, 2), ref, y_true)
    d = eng.describe()
    print(f'    calibration {d["calibration_max_abs_logit_diff_x2m_vs_fp16x2"]:.2e} > {d["threshold"]:.0e} -> {d["form"]}')
    assert d['form'] == 'fp16x2' and d['calibration_max_abs_logit_diff_x2m_vs_fp16x2'] > d['threshold']
    _assert_fp32_mode(r)
    # ... and with the weights back to normal the same engine returns to x2m at its next calibration
    p2 = unet_ref.init_params(dim=2, ncls=2, seed=6, randomize_bn=True)
    eng.recal_every = 1
    eng.load_eval({k: v.cuda() for k, v in p2.items()})
    _forward(eng, x.cuda(), 2, 2)                       # issues the (asynchronous) calibration, still in fp16x2
    assert eng.form == 'fp16x2' and eng.calibrations == 2
    eng.load_eval({k: v.cuda() for k, v in p2.items()})   # adopted at the next weight load
    assert eng.form == 'x2m'
    ref2 = unet_ref.forward_logits(p2, x.float() / 255.0, dim=2)
    _assert_fp32_mode(_compare('auto, normal head again', *_forward(eng, x.cuda(), 2, 2), ref2, y_true))


@pytest.mark.parametrize('gain,want_form', [(3000.0, 'fp16x2_wide'), (3.0e5, 'fp32')])
def test_saturated_forward_is_rerun_in_a_wider_form(gain, want_form, capsys):
    """enc0.conv1 x gain, enc0.conv2 / gain: the same function up to the BatchNorm shifts, but the first activation is ~gain -- beyond
    65504 / 2^6 = 1023 (and, for the larger gain, beyond 65504).  UNet.forward / predict_slice must notice (the range flag) and predict
    again in a form that holds the value: fp16x2 at act_scale 1, then the fp32 mode.  No warning-and-wrong-result."""
    from interactive_unet import predict as P
    p = unet_ref.init_params(dim=2, ncls=2, seed=4, randomize_bn=True)
    p['enc0.conv1.weight'] = p['enc0.conv1.weight'] * gain
    p['enc0.bn1.bias'] = p['enc0.bn1.bias'] * gain
    p['enc0.bn1.running_mean'] = p['enc0.bn1.running_mean'] * gain
    p['enc0.conv2.weight'] = p['enc0.conv2.weight'] / gain
    shape = (96, 128)
    img = _smooth(shape, 41, sigma=4)
    x = torch.tensor(img)[None, None]
    want = unet_ref.forward(p, x.float() / 255.0, dim=2)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=2)
    m = _unet(dim=2)
    m.load_named(p)
    m = m.cuda().eval()
    got = m(x.cuda()).cpu()
    eng = m.engine('eval')
    out = capsys.readouterr().out
    print(out)
    assert 'predicting again' in out
    assert eng.form == want_form, eng.form
    assert not eng.saturated()
    assert (got - want).abs().max().item() <= 2e-4
    r = _compare(f'after the re-run ({want_form})', *_forward(eng, x.cuda(), 2, 2), ref, _labels(img[None, None], 2))
    _assert_fp32_mode(r)
    # the slice entry point (predict.py:16-47) on a fresh module: colours of the oracle's class map
    m2 = _unet(dim=2)
    m2.load_named(p)
    m2 = m2.cuda().eval()
    rgb = P.predict_slice(img, model=m2)
    cls = ref.argmax(1)[0].numpy()
    top2 = torch.topk(ref, 2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1])[0] > 1e-3).numpy()
    assert (rgb[clear] == P.COLORS[1:][cls[clear]]).all()
    assert m2.engine('eval').form == want_form


def test_recalibration_inside_a_training_loop_needs_no_synchronisation():
    """train step -> engine('eval') -> predict, as bench.py's step does: the first calibration blocks, later ones (every `recal_every`
    weight loads) are issued behind the forward and adopted at the next load; the selected form keeps the prediction within tolerance of
    the native fp32 mode throughout."""
    from interactive_unet.train_engine import TrainEngine
    from interactive_unet.engine_f32 import EngineF32
    dev = torch.device('cuda')
    m = _unet(lr=1e-3, dim=2)
    m.reset_parameters(seed=2)
    m = m.to(dev)
    te = TrainEngine(m, lr=1e-3)
    shape, B = (128, 128), 4
    X = torch.tensor(np.stack([_smooth(shape, 300 + i) for i in range(B)])[:, None]).to(dev)
    lab = X > 127
    Y = torch.cat([~lab, lab], 1).to(torch.float16)
    Wt = torch.ones_like(Y)
    vox = shape[0] * shape[1]
    lg = torch.empty((B, 2) + shape, device=dev)
    lg32 = torch.empty_like(lg)
    e32 = EngineF32(dim=2)
    cal_steps = []
    for s in range(40):
        te.train_step(X, Y, Wt, sync=False)
        eng = m.engine('eval')
        before = eng.calibrations
        eng.infer(X, (vox, vox, vox, shape[1], 1), B, 1, shape[0], shape[1], logits=lg)
        if eng.calibrations != before:
            cal_steps.append(s)
        if s % 13 == 0 or s == 39:
            e32.load_eval(m.named_tensors())
            e32.infer(X, (vox, vox, vox, shape[1], 1), B, 1, shape[0], shape[1], logits=lg32)
            err = float((lg - lg32).abs().max())
            print(f'[loop] step {s}: form {eng.form}, |logit - fp32 mode| {err:.2e} at scale {float(lg32.abs().max()):.1f}')
            assert err <= TOL
    assert cal_steps[0] == 0 and len(cal_steps) == 3, cal_steps                 # loads 1, 17, 33
    assert [b - a for a, b in zip(cal_steps, cal_steps[1:])] == [eng.recal_every] * 2
    assert eng.calibration['load'] in (33, 34)
