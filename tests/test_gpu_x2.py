"""fp16x2 split-precision mode (csrc/split16.hip, conv3_v4.hip SPL, interactive_unet/engine_x2.py): the kernels one by one
against float64 CPU references of the same operator on the same fp32 data, then the network against the fp32 oracle
(the north-star gate at the headline sizes is in test_gpu_parity.py).

What each kernel test proves: (a) on small-integer data (every lo word zero, every product and sum exact) the result is
bit-equal -- fragment maps, the chunk -> plane remap of the three virtual parts, halos; (b) on random fp32 data the error
is ~1e-6 of the output scale -- two orders below what one dropped cross term (x_lo w_hi or x_hi w_lo: 2^-12) would leave.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_ref

A = 64.0          # activation scale of the tests (EngineX2 default)


def _nv():
    from interactive_unet import _native as nv
    return nv


def _engine(dim, **kw):
    from interactive_unet.engine_x2 import EngineX2
    return EngineX2(dim=dim, mixed=False, **kw)          # (the 3-D default, cross terms on the fp8 matrix cores: tests/test_gpu_x2m.py)


def _prep_conv(nv, w, bn=None, bias=None, transposed=False):
    """device operator of one layer: (packed f16, oscale, bias) -- what EngineX2.load_eval does."""
    dev = 'cuda'
    w = w.to(dev, torch.float32).contiguous()
    if transposed:
        ci, co = w.shape[:2]
    else:
        co, ci = w.shape[:2]
    taps = int(np.prod(w.shape[2:]))
    nparts = 2 if transposed else 3
    wv = torch.empty(nparts * ci * co * taps, device=dev)
    osc, b = torch.empty(co, device=dev), torch.empty(co, device=dev)
    bnp = [None] * 4 if bn is None else [t.to(dev, torch.float32).contiguous() for t in bn]
    bi = None if bias is None else bias.to(dev, torch.float32).contiguous()
    nv.call('iunet_x2_prep', nv.ptr(w), nv.ptr(wv), nv.ptr(osc), nv.ptr(b), *[nv.ptr(t) for t in bnp], nv.ptr(bi),
            1e-5, A, A, co, ci, taps, 2 if transposed else 0, 0 if transposed else (ci if ci <= 4 else 16 if taps == 27 else 32), nv.stream())
    if transposed:
        dst = torch.empty(2 * ci * co * taps, dtype=torch.float16, device=dev)
        nv.call('iunet_pack_convT', 0, nv.ptr(wv), nv.ptr(dst), 2 * ci, co, taps, nv.stream())
    elif ci <= 4:
        dst = torch.empty(nv.lib().iunet_pack_first_conv_elems(co, 3 * ci, taps), dtype=torch.float16, device=dev)
        nv.call('iunet_pack_first_conv', 0, nv.ptr(wv), None, nv.ptr(dst), co, 3 * ci, taps, nv.stream())
    else:
        pm = nv.lib().iunet_x2_pack_mode(3 if taps == 27 else 2)       # padded K16 order in 3-D, compact (cross-pair step) in 2-D
        dst = torch.empty(nv.pack_conv3_elems(co, 3 * ci, taps, pm), dtype=torch.float16, device=dev)
        nv.call('iunet_pack_conv3', 0, nv.ptr(wv), None, nv.ptr(dst), co, 3 * ci, taps, pm, nv.stream())
    torch.cuda.synchronize()
    return dst, osc, b


def _conv_ref(x, w, dim, bias=None, relu=False):
    y = (F.conv3d if dim == 3 else F.conv2d)(x.double(), w.double(), None if bias is None else bias.double(), padding=1)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize('dim,shape,ci,co,N', [
    (3, (8, 16, 32), 32, 32, 1),
    (3, (6, 10, 20), 64, 32, 2),        # ragged tiles
    (3, (4, 8, 16), 96, 64, 1),         # small-tile variant, two Cout tiles
    (2, (32, 64), 32, 32, 2),
    (2, (24, 40), 64, 64, 1),
    (2, (16, 32), 128, 32, 1),          # 12 virtual steps on the cross-pair order
    (2, (40, 72), 96, 64, 3),           # ragged tiles, several tiles per workgroup
])
def test_conv3_x2(dim, shape, ci, co, N):
    nv, e = _nv(), _engine(dim)
    g = torch.Generator().manual_seed(1)
    sp = (1,) + shape if dim == 2 else shape
    vox = int(np.prod(shape))
    for mode in ('int', 'rand'):
        if mode == 'int':
            x = torch.randint(-3, 4, (N, ci) + shape, generator=g).float() / A           # stored value = the integer: hi only
            w = torch.randint(-2, 3, (co, ci) + (3,) * dim, generator=g).float()
            bias = torch.randint(-4, 5, (co,), generator=g).float()
        else:
            x = torch.rand((N, ci) + shape, generator=g) * 2
            w = torch.randn((co, ci) + (3,) * dim, generator=g) * (2.0 / (ci * 3 ** dim)) ** 0.5
            bias = torch.randn(co, generator=g) * 0.1
        wpk, osc, b = _prep_conv(nv, w, bias=bias)
        xs = e.to_split(x).cuda()
        y = torch.empty(N * 2 * co * vox, dtype=torch.float16, device='cuda')
        nv.call('iunet_x2_conv3_fwd', dim, nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(y), 2 * co * vox, co // 8, nv.ptr(wpk),
                nv.ptr(osc), nv.ptr(b), N, sp[0], sp[1], sp[2], ci, co, 2, nv.stream())
        torch.cuda.synchronize()
        got = e.from_split(y.cpu(), N, co, shape).double()
        want = _conv_ref(x, w, dim, bias, relu=True)
        if mode == 'int':
            assert torch.equal(got, want), (got - want).abs().max()
        else:
            # what went in is the split of x (22 bits), so compare with the conv of exactly that
            xq = e.from_split(e.to_split(x), N, ci, shape)
            want = _conv_ref(xq, w, dim, bias, relu=True)
            err = (got - want).abs().max().item() / want.abs().max().item()
            print(f'[x2 conv {dim}-D {ci}->{co} {shape}] max rel err {err:.2e}')
            assert err < 3e-6, err


@pytest.mark.parametrize('dim,shape,cin', [(3, (8, 16, 32), 1), (3, (6, 10, 20), 3), (2, (32, 64), 1), (2, (24, 40), 4)])
def test_first_conv_x2(dim, shape, cin):
    nv, e = _nv(), _engine(dim, cin=cin)
    g = torch.Generator().manual_seed(2)
    N, co = 2, 32
    sp = (1,) + shape if dim == 2 else shape
    vox = int(np.prod(shape))
    xu = torch.randint(0, 256, (N, cin) + shape, generator=g, dtype=torch.uint8)
    w = torch.randn((co, cin) + (3,) * dim, generator=g) * (2.0 / (cin * 3 ** dim)) ** 0.5
    bn = [0.75 + 0.5 * torch.rand(co, generator=g), 0.1 * torch.randn(co, generator=g), 0.2 * torch.randn(co, generator=g),
          0.5 + torch.rand(co, generator=g)]
    wpk, osc, b = _prep_conv(nv, w, bn=bn)
    y = torch.empty(N * 2 * co * vox, dtype=torch.float16, device='cuda')
    xd = xu.cuda()
    nv.call('iunet_x2_first_conv_fwd', dim, nv.ptr(xd), 2, nv.ll_array((cin * vox, vox, sp[1] * sp[2], sp[2], 1)), nv.ptr(y),
            2 * co * vox, co // 8, nv.ptr(wpk), nv.ptr(osc), nv.ptr(b), A, N, sp[0], sp[1], sp[2], cin, co, 1, nv.stream())
    torch.cuda.synchronize()
    got = e.from_split(y.cpu(), N, co, shape).double()
    wf, bf = unet_ref.fold_bn(w, *bn)
    want = _conv_ref(xu.float() / 255.0, wf, dim, bf, relu=True)
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f'[x2 first conv {dim}-D cin={cin}] max rel err {err:.2e}')
    assert err < 3e-6, err


@pytest.mark.parametrize('dim,shape,ci,co', [(3, (4, 6, 10), 64, 32), (2, (10, 18), 64, 32), (3, (2, 4, 20), 96, 64), (3, (4, 4, 8), 256, 128)])
def test_maxpool_convT_x2(dim, shape, ci, co):
    nv, e = _nv(), _engine(dim)
    g = torch.Generator().manual_seed(3)
    N = 2
    sp = (1,) + shape if dim == 2 else shape
    vox = int(np.prod(shape))
    x = torch.randn((N, ci) + shape, generator=g)
    xs = e.to_split(x).cuda()
    xq = e.from_split(xs.cpu(), N, ci, shape)
    # transposed conv
    w = torch.randn((ci, co) + (2,) * dim, generator=g) * (1.0 / ci) ** 0.5
    bias = torch.randn(co, generator=g) * 0.1
    wpk, osc, b = _prep_conv(nv, w, bias=bias, transposed=True)
    npos = 2 ** dim
    y = torch.empty(N * 2 * co * vox * npos, dtype=torch.float16, device='cuda')
    nv.call('iunet_x2_convT_fwd', dim, nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(y), 2 * co * vox * npos, co // 8, nv.ptr(wpk),
            nv.ptr(osc), nv.ptr(b), N, sp[0], sp[1], sp[2], ci, co, nv.stream())
    torch.cuda.synchronize()
    oshape = tuple(2 * s for s in shape)
    got = e.from_split(y.cpu(), N, co, oshape).double()
    want = (F.conv_transpose3d if dim == 3 else F.conv_transpose2d)(xq.double(), w.double(), bias.double(), stride=2)
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f'[x2 convT {dim}-D] max rel err {err:.2e}')
    assert err < 3e-6, err
    # max-pool: exactly the split values of its source
    if all(s % 2 == 0 for s in shape):
        pshape = tuple(s // 2 for s in shape)
        psp = (1,) + pshape if dim == 2 else pshape
        yp = torch.empty(N * 2 * ci * int(np.prod(pshape)), dtype=torch.float16, device='cuda')
        nv.call('iunet_x2_maxpool_fwd', dim, nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(yp), 2 * ci * int(np.prod(pshape)), ci // 8,
                ci, N, psp[0], psp[1], psp[2], nv.stream())
        torch.cuda.synchronize()
        gotp = e.from_split(yp.cpu(), N, ci, pshape)
        wantp = (F.max_pool3d if dim == 3 else F.max_pool2d)(xq, 2)
        assert torch.equal(gotp, wantp)


def test_fp16_subnormal_lo_words_survive_the_matrix_instruction():
    """Small activations have subnormal lo words (|v * act_scale| < 2^-2); if the MFMA flushed them the error of such a value
    would jump from 2^-24 to 2^-12 relative.  Inputs around 1e-4 (stored 6.4e-3: hi normal, lo deep in the subnormal range)."""
    nv, e = _nv(), _engine(3)
    g = torch.Generator().manual_seed(4)
    shape, ci, co = (4, 8, 16), 32, 32
    vox = int(np.prod(shape))
    x = (torch.rand((1, ci) + shape, generator=g) + 0.5) * 1e-4
    w = torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05
    wpk, osc, b = _prep_conv(nv, w, bias=torch.zeros(co))
    xs = e.to_split(x).cuda()
    y = torch.empty(2 * co * vox, dtype=torch.float16, device='cuda')
    nv.call('iunet_x2_conv3_fwd', 3, nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(y), 2 * co * vox, co // 8, nv.ptr(wpk),
            nv.ptr(osc), nv.ptr(b), 1, shape[0], shape[1], shape[2], ci, co, 1, nv.stream())
    torch.cuda.synchronize()
    got = e.from_split(y.cpu(), 1, co, shape).double()
    xq = e.from_split(e.to_split(x), 1, ci, shape)
    want = _conv_ref(xq, w, 3)
    x_hi = (x * A).to(torch.float16).float() / A                      # what a flushing matrix instruction would have multiplied
    flushed = _conv_ref(x_hi, w, 3)
    err = (got - want).abs().max().item()
    gap = (flushed - want).abs().max().item()
    print(f'[x2 subnormal lo] |got - full| = {err:.2e}, |flushed - full| = {gap:.2e}')
    assert err < 0.05 * gap, (err, gap)


@pytest.mark.parametrize('dim,shape,cin,ncls,in_dtype', [
    (2, (64, 96), 1, 3, torch.uint8),
    (2, (40, 72), 3, 2, torch.float32),        # ragged tiles, 3 input channels
    (3, (16, 32, 48), 1, 3, torch.uint8),
    (3, (8, 24, 40), 2, 4, torch.float16),     # deepest level 1 x 3 x 5: every tile is partial
])
def test_network_small_shapes(dim, shape, cin, ncls, in_dtype):
    from tests.test_gpu_parity import _smooth, _forward, _compare, _assert_fp32_mode, _labels
    p = unet_ref.init_params(dim=dim, cin=cin, ncls=ncls, seed=3, randomize_bn=True)
    N = 2
    img = np.stack([np.stack([_smooth(shape, 10 * i + c) for c in range(cin)]) for i in range(N)])
    x = torch.tensor(img)
    if in_dtype == torch.uint8:
        xd, xf = x.cuda(), x.float() / 255.0
    else:
        xf = (x.float() / 255.0).to(in_dtype).float()
        xd = xf.to(in_dtype).cuda()
    ref = unet_ref.forward_logits(p, xf, dim=dim)
    e = _engine(dim, cin=cin, ncls=ncls)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    r = _compare(f'fp16x2 {dim}-D {shape} cin={cin}', *_forward(e, xd, dim, ncls), ref, _labels(img, ncls))
    _assert_fp32_mode(r)
    assert r['err'] <= 1e-4 * max(1.0, r['scale'])


def test_c5_geometry_in_split_precision():
    """5 levels, base 64, 4 classes (the network of BASELINE.json configs[4]) in fp16x2: 1 024 real = 3 072 virtual input channels at the
    widest decoder conv.  Against the fp32 oracle on a small block."""
    from tests.test_gpu_parity import _smooth, _forward, _compare, _assert_fp32_mode, _labels
    from interactive_unet.engine_x2 import EngineX2
    dim, shape, ncls = 3, (16, 16, 32), 4
    p = unet_ref.init_params(dim=dim, levels=5, base=64, ncls=ncls, seed=4, randomize_bn=True)
    img = np.stack([_smooth(shape, 3 + i) for i in range(1)])[:, None]
    x = torch.tensor(img)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=dim, levels=5)
    e = EngineX2(dim=dim, levels=5, base=64, ncls=ncls, mixed=False)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    r = _compare('fp16x2 3-D 5 levels base 64', *_forward(e, x.cuda(), dim, ncls), ref, _labels(img, ncls))
    _assert_fp32_mode(r)


def test_activations_beyond_the_fp16_range_saturate_finite():
    """act_scale x |activation| above 65504 clamps (split16): the result stays finite, and with a smaller act_scale the same network is
    exact again -- the documented range of the mode (|activation| <= 65504 / act_scale = 1 023 at the default 2^6)."""
    from interactive_unet.engine_x2 import EngineX2
    dim, shape = 2, (32, 32)
    p = unet_ref.init_params(dim=dim, ncls=2, seed=2, randomize_bn=True)
    p['enc0.bn1.weight'] = p['enc0.bn1.weight'] * 4000.0                 # activations of the first layer in the thousands
    x = torch.rand((1, 1) + shape)
    ref = unet_ref.forward_logits(p, x, dim=dim)
    assert torch.isfinite(ref).all()
    out = {}
    for scale in (64.0, 1.0):
        e = EngineX2(dim=dim, act_scale=scale)
        e.load_eval({k: v.cuda() for k, v in p.items()})
        lg = torch.empty((1, 2) + shape, device='cuda')
        e.infer(x.cuda(), (1024, 1024, 1024, 32, 1), 1, 1, 32, 32, logits=lg)
        torch.cuda.synchronize()
        out[scale] = lg.cpu()
        assert torch.isfinite(out[scale]).all()
        assert e.saturated() == (scale == 64.0), (scale, e.max_stored())          # the diagnostic sees the clamp
    assert (out[1.0] - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
