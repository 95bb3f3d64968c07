"""Handle-level C ABI (csrc/net.hip, include/iunet.h "handle level"): the whole forward as one C call.

CPU part: the host-side logic (canonical parameter order = the module's state_dict, buffer sizes, refusals).  GPU part: the C++
launch graph gives the same bits as the Python-sequenced engines (same kernels, same operators) and meets the north-star tolerance
against the CPU oracle in its split-precision mode."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import unet_ref


def _nv():
    from interactive_unet import _native as nv
    nv.lib()
    return nv


def _create(nv, dim, levels, base, cin, ncls, mode, act_scale=0.0):
    h = ctypes.c_void_p()
    nv.call('iunet_net_create', dim, levels, base, cin, ncls, mode, act_scale, ctypes.byref(h))
    return h


def _layout(nv, h):
    out = []
    for i in range(nv.lib().iunet_net_num_tensors(h)):
        name = ctypes.create_string_buffer(64)
        off, n = ctypes.c_longlong(), ctypes.c_longlong()
        nv.call('iunet_net_param', h, i, name, 64, ctypes.byref(off), ctypes.byref(n))
        out.append((name.value.decode(), off.value, n.value))
    return out


@pytest.mark.parametrize('dim,levels,base,cin,ncls', [(2, 4, 32, 1, 2), (3, 4, 32, 1, 2), (3, 5, 64, 2, 4)])
def test_parameter_layout_is_the_state_dict_order(dim, levels, base, cin, ncls):
    from interactive_unet.unet import param_shapes
    nv = _nv()
    h = _create(nv, dim, levels, base, cin, ncls, 2)
    lay = _layout(nv, h)
    shapes = param_shapes(dim, levels, base, cin, ncls)
    assert [n for n, _, _ in lay] == list(shapes)
    off = 0
    for (name, o, n), shp in zip(lay, shapes.values()):
        assert o == off and n == int(np.prod(shp)), name
        off += n
    assert nv.lib().iunet_net_num_params(h) == off
    assert nv.lib().iunet_net_packed_bytes(h) > 2 * sum(int(np.prod(s)) for k, s in shapes.items() if k.endswith('conv2.weight'))
    f = 2 ** (levels - 1)
    good = (1, 4 * f, 4 * f) if dim == 2 else (2 * f, 2 * f, 4 * f)
    assert nv.lib().iunet_net_workspace_bytes(h, 2, *good) > 0
    assert nv.lib().iunet_net_workspace_bytes(h, 2, good[0], good[1] + 1, good[2]) == 0          # not divisible: refused
    nv.lib().iunet_net_destroy(h)


def test_refusals():
    nv = _nv()
    h = ctypes.c_void_p()
    l = nv.lib()
    assert l.iunet_net_create(4, 4, 32, 1, 2, 2, 0.0, ctypes.byref(h)) < 0 and b'dim' in l.iunet_last_error()
    assert l.iunet_net_create(2, 4, 48, 1, 2, 2, 0.0, ctypes.byref(h)) < 0 and b'base' in l.iunet_last_error()
    assert l.iunet_net_create(2, 4, 32, 1, 2, 4, 0.0, ctypes.byref(h)) < 0 and b'mode' in l.iunet_last_error()
    h = _create(nv, 2, 4, 32, 1, 2, 2)
    st = nv.ll_array((1, 1, 1, 1, 1))
    assert l.iunet_net_forward(h, ctypes.c_void_p(8), 2, st, 1, 1, 64, 64, ctypes.c_void_p(8), None, None, ctypes.c_void_p(8), None, 1.0, 0, None) < 0
    assert b'iunet_net_load' in l.iunet_last_error()                      # refused before any launch
    l.iunet_net_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize('dim,shape,mode', [(2, (64, 96), 2), (2, (64, 96), 3), (3, (16, 32, 48), 2), (3, (16, 32, 48), 3), (2, (64, 96), 0), (3, (16, 32, 48), 1)])
def test_c_graph_matches_the_python_engines_and_the_oracle(dim, shape, mode):
    from interactive_unet.engine import Engine
    from interactive_unet.engine_x2 import EngineX2
    nv = _nv()
    ncls, N = 3, 2
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=11, randomize_bn=True)
    h = _create(nv, dim, 4, 32, 1, ncls, mode)
    flat = torch.empty(nv.lib().iunet_net_num_params(h), device='cuda')
    for name, off, n in _layout(nv, h):
        flat[off:off + n] = p[name].reshape(-1).cuda()
    packed = torch.empty(nv.lib().iunet_net_packed_bytes(h), dtype=torch.uint8, device='cuda')
    nv.call('iunet_net_load', h, nv.ptr(flat), nv.ptr(packed), nv.stream())
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    ws = torch.zeros(nv.lib().iunet_net_workspace_bytes(h, N, D, H, W), dtype=torch.uint8, device='cuda')      # (mode 3: its first int is the range flag)
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.integers(0, 256, (N, 1) + shape, dtype=np.uint8)).cuda()
    logits = torch.empty((N, ncls) + shape, device='cuda')
    probs = torch.empty((N, ncls) + shape, device='cuda')
    cls = torch.empty((N, vox), dtype=torch.uint8, device='cuda')
    st = nv.ll_array((vox, vox, H * W, W, 1))
    os_ = nv.ll_array((ncls * vox, vox, H * W, W, 1))
    nv.call('iunet_net_forward', h, nv.ptr(x), 2, st, N, D, H, W, nv.ptr(ws), nv.ptr(logits), nv.ptr(probs), nv.ptr(cls), os_, 1.0, 0, nv.stream())
    cls2 = torch.empty_like(cls)
    nv.call('iunet_net_forward_argmax', h, nv.ptr(x), nv.ptr(cls2), N, D, H, W, nv.ptr(ws), nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(cls, cls2)
    e = EngineX2(dim=dim, ncls=ncls, mixed=(mode == 3)) if mode >= 2 else Engine(dim=dim, ncls=ncls, act_dtype=(torch.float16, torch.bfloat16)[mode])
    e.load_eval({k: v.cuda() for k, v in p.items()})
    lg2, cl2 = torch.empty_like(logits), torch.empty_like(cls)
    e.use_graph = False                                   # the engine's own launch sequence
    e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, logits=lg2, cls=cl2)
    torch.cuda.synchronize()
    assert torch.equal(logits, lg2) and torch.equal(cls, cl2), 'the C++ graph and the Python-sequenced engine launch the same kernels'
    # ... and the engine's default route IS that graph (net_graph.NetGraph behind Engine.infer): strided 2.5-D style output included
    e.use_graph = True
    e._g_fwd = 1                                          # (the handle is loaded at the second forward on the same parameters)
    lg3, pr3, pr2 = torch.empty_like(logits), torch.full_like(probs, 0.25), torch.full_like(probs, 0.25)
    e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, logits=lg3, probs=pr3, accumulate=True, divisor=3.0)
    from interactive_unet import net_graph
    assert not net_graph.ENABLED or (e._g is not None and e._g.loaded)       # (IUNET_PY_GRAPH=1 keeps the Python sequence)
    e.use_graph = False
    e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, probs=pr2, accumulate=True, divisor=3.0)
    torch.cuda.synchronize()
    assert torch.equal(lg3, lg2) and torch.equal(pr3, pr2)
    ref = unet_ref.forward_logits(p, x.cpu().float() / 255.0, dim=dim)
    err = (logits.cpu() - ref).abs().max().item()
    print(f'[net handle {dim}-D mode {mode}] max |logit - CPU fp32 oracle| = {err:.2e}')
    if mode == 2:
        assert err <= 1e-3 and torch.equal(cls.cpu().long().reshape(N, *shape), ref.argmax(1))
    if mode == 3:
        # (this input is white noise: many near-ties.  The class map is equal wherever the oracle's own margin exceeds twice the error)
        top2 = torch.topk(ref, 2, dim=1).values
        mism = cls.cpu().long().reshape(N, *shape) != ref.argmax(1)
        assert err <= 1e-3 and (not mism.any() or (top2[:, 0] - top2[:, 1])[mism].max().item() <= 2 * err)
        assert int(ws[:4].view(torch.int32).item()) == 0          # no activation saturated
    nv.lib().iunet_net_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize('dim,shape', [(2, (64, 96)), (3, (16, 32, 48))])
def test_groupnorm_network_through_the_handle(dim, shape):
    """iunet_net_create_ex(norm = 1): the GroupNorm(8) network in split precision (mode 2) sequenced in C++ -- the same launches as
    engine_x2.EngineX2(norm='group') sequences from Python (torch.equal), within 1e-3 of oracle/unet_ref.forward_logits(norm='group'),
    and the engine's own default route from its second forward on."""
    from interactive_unet.engine_x2 import EngineX2
    from interactive_unet import net_graph
    nv = _nv()
    l = nv.lib()
    ncls, N = 3, 2
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=12, randomize_bn=True)
    h = ctypes.c_void_p()
    assert l.iunet_net_create_ex(dim, 4, 32, 1, ncls, 0, 0.0, 1, 8, ctypes.byref(h)) < 0 and b'GroupNorm' in l.iunet_last_error()      # 16-bit modes: BatchNorm only
    nv.call('iunet_net_create_ex', dim, 4, 32, 1, ncls, 2, 0.0, 1, 8, ctypes.byref(h))
    flat = torch.empty(l.iunet_net_num_params(h), device='cuda')
    for name, off, n in _layout(nv, h):
        flat[off:off + n] = p[name].reshape(-1).cuda()
    packed = torch.empty(l.iunet_net_packed_bytes(h), dtype=torch.uint8, device='cuda')
    nv.call('iunet_net_load', h, nv.ptr(flat), nv.ptr(packed), nv.stream())
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    ws = torch.zeros(l.iunet_net_workspace_bytes(h, N, D, H, W), dtype=torch.uint8, device='cuda')
    rng = np.random.default_rng(5)
    x = torch.tensor(rng.integers(0, 256, (N, 1) + shape, dtype=np.uint8)).cuda()
    logits = torch.empty((N, ncls) + shape, device='cuda')
    cls = torch.empty((N, vox), dtype=torch.uint8, device='cuda')
    st = nv.ll_array((vox, vox, H * W, W, 1))
    os_ = nv.ll_array((ncls * vox, vox, H * W, W, 1))
    nv.call('iunet_net_forward', h, nv.ptr(x), 2, st, N, D, H, W, nv.ptr(ws), nv.ptr(logits), None, nv.ptr(cls), os_, 1.0, 0, nv.stream())
    e = EngineX2(dim=dim, ncls=ncls, norm='group', groups=8, mixed=False)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    lg2, cl2, lg3 = torch.empty_like(logits), torch.empty_like(cls), torch.empty_like(logits)
    e.use_graph = False
    e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, logits=lg2, cls=cl2)
    e.use_graph = True
    e._g_fwd = 1
    e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, logits=lg3)
    torch.cuda.synchronize()
    assert torch.equal(logits, lg2) and torch.equal(cls, cl2) and torch.equal(lg3, lg2)
    assert not net_graph.ENABLED or (e._g is not None and e._g.loaded)
    ref = unet_ref.forward_logits(p, x.cpu().float() / 255.0, dim=dim, norm='group', groups=8)
    err = (logits.cpu() - ref).abs().max().item()
    print(f'[net handle {dim}-D GroupNorm, mode 2] max |logit - CPU fp32 oracle| = {err:.2e}')
    assert err <= 1e-4
    assert int(ws[:4].view(torch.int32).item()) == 0
    l.iunet_net_destroy(h)
