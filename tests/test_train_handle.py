"""Handle-level C ABI of the training step (csrc/train_net.hip, include/iunet.h "the TRAINING step as one call"): unet.py:88-102 + backward
+ AdamW (unet.py:71-73) + the operator re-pack sequenced in C++.  It must be the launches interactive_unet/train_engine.py sequences
from Python, bit for bit -- parameters, AdamW moments, BatchNorm running statistics, metrics and the device training state after several
steps, in 2-D and 3-D, fp16 (dynamic loss scale, overflow back-off on the device) and bf16; and a step driven through the bare C ABI
(no TrainEngine) gives the same bits again."""
import ctypes
import warnings

import numpy as np
import pytest
import torch

from oracle import unet_ref


def _nv():
    from interactive_unet import _native as nv
    return nv


def test_train_handle_argument_checks_without_gpu():
    nv = _nv()
    l = nv.lib()
    h = ctypes.c_void_p()
    assert l.iunet_train_create(4, 4, 32, 1, 2, 0, 6, ctypes.byref(h)) < 0 and b'dim' in l.iunet_last_error()
    assert l.iunet_train_create(2, 4, 48, 1, 2, 0, 6, ctypes.byref(h)) < 0 and b'base' in l.iunet_last_error()
    assert l.iunet_train_create(2, 4, 32, 1, 2, 2, 6, ctypes.byref(h)) < 0 and b'dtype' in l.iunet_last_error()
    assert l.iunet_train_create(2, 4, 32, 1, 2, 0, 9, ctypes.byref(h)) < 0 and b'loss' in l.iunet_last_error()
    assert l.iunet_train_create(2, 4, 32, 1, 2, 0, 6, ctypes.byref(h)) == 0
    # the flat layout = the trainable tensors of the canonical network in state_dict order (no running statistics)
    names = []
    for i in range(l.iunet_train_num_tensors(h)):
        name = ctypes.create_string_buffer(96)
        off, n = ctypes.c_longlong(), ctypes.c_longlong()
        assert l.iunet_train_param(h, i, name, 96, ctypes.byref(off), ctypes.byref(n)) == 0
        names.append((name.value.decode(), off.value, n.value))
    shapes = unet_ref.param_shapes(dim=2)
    want = [k for k in shapes if not unet_ref.is_buffer(k)]
    assert [n for n, _, _ in names] == want
    off = 0
    for n, o, c in names:
        assert o == off and c == int(np.prod(shapes[n]))
        off += c
    assert l.iunet_train_num_params(h) == off and l.iunet_train_num_bn(h) == 14
    assert l.iunet_train_workspace_bytes(h, 2, 1, 60, 64) == 0 and l.iunet_train_workspace_bytes(h, 2, 1, 64, 64) > 0
    st = nv.ll_array((1, 1, 1, 1, 1))
    assert l.iunet_train_step(h, ctypes.c_void_p(8), 2, st, ctypes.c_void_p(8), None, 1, 1, 1, 64, 64, ctypes.c_void_p(8), 1e-4, 0.9, 0.999,
                              1e-8, 1e-2, None, None) < 0
    assert b'iunet_train_bind' in l.iunet_last_error()                      # refused before any launch
    l.iunet_train_destroy(h)


def _model(dim, dtype, seed=1, norm='batch'):
    from interactive_unet.unet import UNet
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(lr=1e-3, num_classes=2, dim=dim, pretrained=False, act_dtype=dtype, norm=norm)
    m.load_named(unet_ref.init_params(dim=dim, ncls=2, seed=seed))
    return m.cuda()


def _batch(seed, N, shape, scale=1.0):
    rng = np.random.default_rng(seed)
    X = torch.tensor(rng.random((N, 1) + shape, dtype=np.float32)) * scale
    lab = X[:, 0] > 0.5 * scale
    y = torch.stack([~lab, lab], 1).float()
    w = torch.tensor((rng.random((N, 1) + shape) > 0.2).astype(np.float32)).expand(N, 2, *shape).contiguous()
    return X.cuda(), (y * w).to(torch.float16).cuda(), w.to(torch.float16).cuda()


def _same(te_a, te_b, tag):
    for name in ('flat', 'm', 'v', 'state'):
        a, b = getattr(te_a, name), getattr(te_b, name)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f'{tag}: {name} differs (max {float((a - b).abs().max())})'
    for n in te_a.model._names:
        if unet_ref.is_buffer(n):
            assert torch.equal(te_a.model.tensor(n).view(torch.int32), te_b.model.tensor(n).view(torch.int32)), f'{tag}: {n} differs'


@pytest.mark.gpu
@pytest.mark.parametrize('dim,shape,N,dtype', [(2, (64, 96), 2, 'fp16'), (2, (64, 96), 3, 'bf16'), (3, (16, 32, 32), 2, 'bf16'), (3, (16, 16, 32), 1, 'fp16')])
def test_c_sequenced_step_is_the_python_sequenced_step(dim, shape, N, dtype):
    from interactive_unet.train_engine import TrainEngine
    a, b = _model(dim, dtype), _model(dim, dtype)
    te_a = TrainEngine(a, lr=1e-3, loss_kind='mcc_ce')
    te_b = TrainEngine(b, lr=1e-3, loss_kind='mcc_ce')
    te_a.use_handle = False                                    # every step sequenced from Python
    te_b._steps_seen = 1                                       # (the handle takes over at the second step: here from the first)
    for step in range(4):
        # step 2 (fp16): a loss scale that overflows the fp16 gradient -- the step is skipped and the scale halved, on the device, by both
        batch = _batch(step, N, shape)
        if dtype == 'fp16' and step == 2:
            te_a.loss_scale = te_b.loss_scale = 2.0 ** 30
            before = te_a.flat.clone()
        ra = te_a.train_step(*batch)
        rb = te_b.train_step(*batch)
        assert getattr(te_b, '_h', None) is not None and getattr(te_a, '_h', None) is None
        assert ra == rb, (step, ra, rb)
        _same(te_a, te_b, f'step {step}')
        if dtype == 'fp16' and step == 2:
            assert torch.equal(before, te_a.flat) and not te_a.last_step_ok and te_a.loss_scale == 2.0 ** 29 and te_a.step_count == 2
            te_a.loss_scale = te_b.loss_scale = 1024.0
    if dtype == 'fp16':
        assert te_a.loss_scale == te_b.loss_scale and te_a.step_count == 3
        print(f'[train handle {dim}-D fp16] loss scale after 4 steps {te_b.loss_scale}, steps counted {te_b.step_count}')
    # the two sequences share the weights, the moments and the state: a Python-sequenced step behind C++-sequenced ones continues them
    te_b.use_handle = False
    batch = _batch(9, N, shape)
    ra, rb = te_a.train_step(*batch), te_b.train_step(*batch)
    assert ra == rb
    _same(te_a, te_b, 'python step behind handle steps')
    # ... and the module predicts with the updated weights
    x = batch[0]
    assert torch.equal(a(x), b(x))


@pytest.mark.gpu
def test_training_step_through_the_bare_c_abi():
    """No TrainEngine: a handle, caller-owned device vectors filled from the oracle's initialisation, three iunet_train_step calls --
    against the Python-sequenced engine on a twin module."""
    from interactive_unet.train_engine import TrainEngine
    nv = _nv()
    l = nv.lib()
    dim, shape, N = 2, (64, 64), 2
    p = unet_ref.init_params(dim=dim, ncls=2, seed=1)
    h = ctypes.c_void_p()
    nv.call('iunet_train_create', dim, 4, 32, 1, 2, 0, 6, ctypes.byref(h))
    n = l.iunet_train_num_params(h)
    flat = torch.empty(n, device='cuda')
    for i in range(l.iunet_train_num_tensors(h)):
        name = ctypes.create_string_buffer(96)
        off, cnt = ctypes.c_longlong(), ctypes.c_longlong()
        nv.call('iunet_train_param', h, i, name, 96, ctypes.byref(off), ctypes.byref(cnt))
        flat[off.value:off.value + cnt.value] = p[name.value.decode()].reshape(-1).cuda()
    grad, m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    bn_names = [f'{pre}.bn{j}' for pre in [f'enc{i}' for i in range(4)] + [f'dec{i}' for i in (2, 1, 0)] for j in (1, 2)]
    running = []
    for b in bn_names:
        running += [p[b + '.running_mean'].clone().cuda(), p[b + '.running_var'].clone().cuda()]
    arr = (ctypes.c_void_p * len(running))(*[t.data_ptr() for t in running])
    state = torch.zeros(8, device='cuda')
    nv.call('iunet_train_state_init', nv.ptr(state), 1024.0, 1, nv.stream())
    packed = torch.empty(l.iunet_train_packed_bytes(h), dtype=torch.uint8, device='cuda')
    nv.call('iunet_train_bind', h, nv.ptr(flat), nv.ptr(grad), nv.ptr(m), nv.ptr(v), arr, nv.ptr(packed), nv.ptr(state), nv.stream())
    ws = torch.empty(l.iunet_train_workspace_bytes(h, N, 1, *shape), dtype=torch.uint8, device='cuda')
    out4 = torch.empty(4, device='cuda')
    twin = _model(dim, 'fp16')
    te = TrainEngine(twin, lr=1e-3, loss_kind='mcc_ce')
    te.use_handle = False
    vox = shape[0] * shape[1]
    for step in range(3):
        X, y, w = _batch(step, N, shape)
        nv.call('iunet_train_step', h, nv.ptr(X), 0, nv.ll_array((vox, vox, vox, shape[1], 1)), nv.ptr(y), nv.ptr(w), 1, N, 1, shape[0], shape[1],
                nv.ptr(ws), 1e-3, 0.9, 0.999, 1e-8, 1e-2, nv.ptr(out4), nv.stream())
        row = te.train_step(X, y, w)
        assert out4.tolist() == [row[k] for k in ('Loss', 'Dice', 'IoU', 'MCC')]
    torch.cuda.synchronize()
    assert torch.equal(flat, te.flat) and torch.equal(m, te.m) and torch.equal(v, te.v) and torch.equal(state.view(torch.int32), te.state.view(torch.int32))
    for b, (rm, rv) in zip(bn_names, zip(running[0::2], running[1::2])):
        assert torch.equal(rm, twin.tensor(b + '.running_mean')) and torch.equal(rv, twin.tensor(b + '.running_var'))
    l.iunet_train_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize('dim,shape,N,dtype', [(2, (64, 96), 2, 'fp16'), (3, (16, 32, 32), 2, 'bf16')])
def test_c_sequenced_validation_step_is_the_python_sequenced_one(dim, shape, N, dtype, monkeypatch):
    """validation_step (unet.py:104-116) as ONE C call (iunet_net_eval_step: the prediction handle's eval-mode forward + the fused head /
    loss kernel) against the Python sequence (engine forward to the head's input + iunet_head_loss_fwd): the same four numbers, bit for bit."""
    from interactive_unet.train_engine import TrainEngine
    m = _model(dim, dtype)
    te = TrainEngine(m, lr=1e-3, loss_kind='mcc_ce')
    for step in range(2):
        te.train_step(*_batch(step, N, shape))
    batch = _batch(7, N, shape)
    monkeypatch.setenv('IUNET_PY_EVAL', '1')
    want = te.eval_step(*batch)
    monkeypatch.delenv('IUNET_PY_EVAL')
    first = te.eval_step(*batch)            # first forward on these weights: still the Python sequence
    got = te.eval_step(*batch)              # second: the handle
    eng = te._eval_engine()
    from interactive_unet import net_graph
    assert not net_graph.ENABLED or (eng._g is not None and eng._g.loaded)
    assert want == first == got, (want, first, got)
    assert 0.0 < got['Loss'] < 10.0 and 0.0 <= got['Dice'] <= 1.0
    # the device-tensor form (trainer.train_model collects these per validation batch)
    t = te.eval_step(*batch, sync=False)
    assert t.tolist() == [got[k] for k in ('Loss', 'Dice', 'IoU', 'MCC')]


@pytest.mark.gpu
@pytest.mark.parametrize('dim,shape,N,dtype', [(2, (64, 96), 2, 'fp16'), (3, (16, 32, 32), 2, 'bf16')])
def test_c_sequenced_groupnorm_step_is_the_python_sequenced_step(dim, shape, N, dtype):
    """iunet_train_create_ex(norm = 1): the GroupNorm(8) training step sequenced in C++ against train_engine.TrainEngine's Python sequence on a
    twin module -- the same launches: losses, parameters, moments and state bit for bit over four steps."""
    from interactive_unet.train_engine import TrainEngine
    a, b = _model(dim, dtype, norm='group'), _model(dim, dtype, norm='group')
    te_a = TrainEngine(a, lr=1e-3, loss_kind='mcc_ce')
    te_b = TrainEngine(b, lr=1e-3, loss_kind='mcc_ce')
    assert te_a.gn and te_b.gn
    te_a.use_handle = False
    te_b._steps_seen = 1
    for step in range(4):
        batch = _batch(step, N, shape)
        ra, rb = te_a.train_step(*batch), te_b.train_step(*batch)
        assert getattr(te_b, '_h', None) is not None and getattr(te_a, '_h', None) is None
        assert ra == rb, (step, ra, rb)
        _same(te_a, te_b, f'GroupNorm step {step}')
    assert ra['Loss'] < 10.0
