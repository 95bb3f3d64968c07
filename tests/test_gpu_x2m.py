"""fp16x2 with the cross terms on the fp8 matrix cores ("x2m": csrc/conv3_x2m.hip, EngineX2(mixed=True)): the 3x3x3 stage conv of
the split-precision forward as x_hi w_hi on v_mfma_f32_16x16x32_f16 + [x_lo8 | x_hi8] [w_hi8 | w_lo8] on v_mfma_f32_16x16x128_f8f6f4.

Kernel level: (a) data on which every operand of BOTH steps is a small integer -- hi planes, hand-made m8 planes, operator entries of the
form 16 a + b / 256 so that w_hi8 = a and w_lo8 = b exactly -- must give the bit-exact conv (lane maps of the two instructions, operator
orders, the pairing of the powers of two, halos, ragged tiles); (b) on random fp32 data the result is within ~2^-14 of the exact conv
of the split input (the fp16 mode leaves 2^-11; a dropped cross term 2^-12).  Network level: logits against the fp32 oracle.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_ref

A = 64.0


def _nv():
    from interactive_unet import _native as nv
    return nv


def _blocked(u, c_blk):
    """[N, C, D, H, W] -> [N, C / c_blk, D, H, W, c_blk] contiguous (channel-blocked planes)."""
    N, C = u.shape[:2]
    sp = u.shape[2:]
    return u.reshape(N, C // c_blk, c_blk, *sp).permute(0, 1, 3, 4, 5, 2).contiguous()


def _e4m3(t):
    return torch.from_numpy(unet_ref.round_e4m3(np.clip(t.numpy().astype(np.float32), -448, 448)))


def _m8_planes(lo8_vals):
    """[N, C, D, H, W] e4m3-representable values -> the lo8 planes [N][C / 16][D][H][W][16] of a tensor, in a sample slot of 2 C voxels bytes (the
    tests keep their buffers at the size of the two-plane format they were written for: a sample stride is just a stride).  The other half of
    the fp8 step's operand, hi8 = e4m3(hi / 256), is a function of the hi words: the conv makes it in LDS."""
    N, C = lo8_vals.shape[:2]
    lo = _blocked(lo8_vals, 16)
    both = torch.cat([lo, torch.zeros_like(lo)], 1)
    return both.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


def _m8_unpack(b, N, C, sp):
    """lo8 values [N, C, *sp] of a buffer whose samples are 2 C voxels bytes apart (the lo8 planes sit at the start of a sample slot)"""
    t = b.view(torch.float8_e4m3fn).float().reshape(N, 2 * C // 16, *sp, 16)[:, :C // 16]
    return t.permute(0, 1, 5, 2, 3, 4).reshape(N, C, *sp)




def _prep(nv, w, bn=None, act_out=A):
    """operators of one stage conv in the x2m form; w [Cout][Cin][3][3][3] (3-D) or [Cout][Cin][3][3] (2-D)"""
    co, ci = w.shape[:2]
    nd = w.dim() - 2
    taps = 3 ** nd
    dev = 'cuda'
    w = w.to(dev, torch.float32).contiguous()
    whi = torch.empty(co * ci * taps, device=dev)
    w8 = torch.zeros(nv.lib().iunet_x2m_w8_bytes_nd(nd, co, ci), dtype=torch.uint8, device=dev)
    osc, b = torch.empty(co, device=dev), torch.empty(co, device=dev)
    bnp = [None] * 4 if bn is None else [t.to(dev, torch.float32).contiguous() for t in bn]
    nv.call('iunet_x2m_prep_nd', nd, nv.ptr(w), nv.ptr(whi), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), *[nv.ptr(t) for t in bnp], 1e-5, A, act_out, co, ci, nv.stream())
    pm = 2 if nd == 3 else 6                              # padded K16 order / the cross-pair order
    w16 = torch.empty(nv.pack_conv3_elems(co, ci, taps, pm), dtype=torch.float16, device=dev)
    nv.call('iunet_pack_conv3', 0, nv.ptr(whi), None, nv.ptr(w16), co, ci, taps, pm, nv.stream())
    torch.cuda.synchronize()
    return w16, w8, osc, b, whi.reshape(w.shape).cpu()


def _run(nv, xhi, x8, w16, w8, osc, bias, N, shape, ci, co, epi, y_lo=True, x_ss=None, nd=3):
    """shape: (D, H, W); nd = 2 needs D == 1"""
    vox = int(np.prod(shape))
    x_ss = ci * vox if x_ss is None else x_ss
    y = torch.zeros(N * 2 * co * vox, dtype=torch.float16, device='cuda')
    y8 = torch.zeros(N * 2 * co * vox, dtype=torch.uint8, device='cuda')
    sat = torch.zeros(1, dtype=torch.int32, device='cuda')
    nv.call('iunet_x2m_conv_fwd', nd, nv.ptr(xhi), x_ss, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), 2 * co * vox, co // 8 if y_lo else -1,
            nv.ptr(y8), 2 * co * vox, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, shape[0], shape[1], shape[2], ci, co, epi,
            nv.ptr(sat), nv.stream())
    torch.cuda.synchronize()
    t = y.cpu().reshape(N, 2, co // 8, *shape, 8).float()
    un = lambda u: u.permute(0, 1, 5, 2, 3, 4).reshape(N, co, *shape)
    return un(t[:, 0]), un(t[:, 1]), y8.cpu(), int(sat.item())


@pytest.mark.parametrize('shape,ci,co,N', [
    ((8, 16, 32), 32, 32, 1),
    ((6, 10, 20), 64, 32, 2),          # ragged tiles
    ((4, 8, 16), 96, 64, 1),           # the half-size tile, two Cout tiles, six chunk pairs
    ((12, 24, 48), 32, 32, 1),         # several tiles per workgroup
])
def test_conv3_x2m_exact_integers(shape, ci, co, N):
    nv = _nv()
    g = torch.Generator().manual_seed(11)
    # operator entries 16 a + b / 256: |a| in {36, 40, .., 60} (hi = 16 a exactly: |b| / 256 stays under half an fp16 ulp of it; a is an e4m3 value), b in 16 x {-3 .. 3} (w_lo8 = b)
    a = (torch.randint(9, 16, (co, ci, 3, 3, 3), generator=g) * 4).float() * (torch.randint(0, 2, (co, ci, 3, 3, 3), generator=g) * 2 - 1).float()
    b = (torch.randint(-3, 4, (co, ci, 3, 3, 3), generator=g) * 16).float()
    w = 16.0 * a + b / 256.0
    w16, w8, osc, bias, whi = _prep(nv, w, act_out=1.0)                      # accumulator / 64: the sums stay below the fp16 range
    assert torch.equal(whi, 16.0 * a)                                        # the row scale is 1 (max |w| in [2^9, 2^10))
    assert torch.equal(osc.cpu(), torch.full((co,), 1.0 / 64)) and torch.equal(bias.cpu(), torch.zeros(co))
    # hi planes: sparse -1 / 0 / 1, so that hi8 = e4m3(hi / 256) = hi / 256 exactly (a subnormal of e4m3) and the sums stay small enough for
    # the fractions of the x_hi8 w_lo8 term (multiples of 2^-4: w_lo8 = b is a multiple of 16) beside the integers of the other two
    X = (torch.randint(-1, 2, (N, ci) + shape, generator=g) * (torch.rand((N, ci) + shape, generator=g) < 0.25)).float()
    L8 = torch.randint(-4, 5, (N, ci) + shape, generator=g).float()           # lo8 plane values
    xhi = _blocked(X, 8).to(torch.float16).cuda()
    x8 = _m8_planes(L8).cuda()
    hi, lo, y8, sat = _run(nv, xhi, x8, w16, w8, osc, bias, N, shape, ci, co, 0)
    want = (F.conv3d(X.double(), (16.0 * a).double(), padding=1) + F.conv3d(L8.double(), a.double(), padding=1)
            + F.conv3d(X.double() / 256.0, b.double(), padding=1))
    assert want.abs().max() < 2 ** 18                                        # + 4 fraction bits = 22: hi + lo hold acc / 64 exactly
    got = (hi.double() + lo.double()) * 64.0
    assert torch.equal(got, want), (got - want).abs().max()
    # the lo8 planes of the output: e4m3 of (v - hi) * 16
    v = (want / 64.0).float()
    h = v.to(torch.float16).float()
    lo8 = _m8_unpack(y8, N, co, shape)
    assert torch.equal(hi, h)
    assert torch.equal(lo8, _e4m3((v - h) * 16.0))
    assert sat == 0


@pytest.mark.parametrize('shape,ci,co,N', [
    ((32, 64), 32, 32, 2),
    ((24, 40), 64, 64, 1),            # ragged tiles, two Cout tiles, two chunk pairs
    ((16, 16), 128, 32, 5),           # images smaller than a tile (slot groups), four chunk pairs
    ((40, 72), 96, 64, 3),            # ragged, several tiles per workgroup
])
def test_conv2_x2m_exact_integers(shape, ci, co, N):
    """The 2-D form (3 x 3 filters, 16 x 32 tiles, a step = 32 channels: cross-pair order for the 16-bit step, four taps per K = 128
    instruction + tap 8 on the K = 32 instruction for the fp8 step) on data that makes every operand of both steps an exact integer."""
    nv = _nv()
    g = torch.Generator().manual_seed(31)
    a = (torch.randint(9, 16, (co, ci, 3, 3), generator=g) * 4).float() * (torch.randint(0, 2, (co, ci, 3, 3), generator=g) * 2 - 1).float()
    b = (torch.randint(-3, 4, (co, ci, 3, 3), generator=g) * 16).float()
    w = 16.0 * a + b / 256.0
    w16, w8, osc, bias, whi = _prep(nv, w, act_out=1.0)
    assert torch.equal(whi, 16.0 * a) and torch.equal(osc.cpu(), torch.full((co,), 1.0 / 64))
    X = (torch.randint(-1, 2, (N, ci) + shape, generator=g) * (torch.rand((N, ci) + shape, generator=g) < 0.25)).float()      # (see the 3-D test)
    L8 = torch.randint(-4, 5, (N, ci) + shape, generator=g).float()
    sp = (1,) + shape
    xhi = _blocked(X.reshape(N, ci, *sp), 8).to(torch.float16).cuda()
    x8 = _m8_planes(L8.reshape(N, ci, *sp)).cuda()
    hi, lo, y8, sat = _run(nv, xhi, x8, w16, w8, osc, bias, N, sp, ci, co, 0, nd=2)
    want = (F.conv2d(X.double(), (16.0 * a).double(), padding=1) + F.conv2d(L8.double(), a.double(), padding=1)
            + F.conv2d(X.double() / 256.0, b.double(), padding=1)).reshape(N, co, *sp)
    assert want.abs().max() < 2 ** 18
    got = (hi.double() + lo.double()) * 64.0
    assert torch.equal(got, want), (got - want).abs().max()
    v = (want / 64.0).float()
    h = v.to(torch.float16).float()
    lo8 = _m8_unpack(y8, N, co, sp)
    assert torch.equal(hi, h) and torch.equal(lo8, _e4m3((v - h) * 16.0)) and sat == 0


@pytest.mark.parametrize('shape,ci,co,N', [((32, 64), 32, 32, 2), ((24, 40), 64, 64, 1), ((16, 32), 256, 32, 1)])
def test_conv2_x2m_random(shape, ci, co, N):
    nv = _nv()
    g = torch.Generator().manual_seed(32)
    x = torch.rand((N, ci) + shape, generator=g) * 2
    w = torch.randn((co, ci, 3, 3), generator=g) * (2.0 / (ci * 9)) ** 0.5
    bn = [0.75 + 0.5 * torch.rand(co, generator=g), 0.1 * torch.randn(co, generator=g), 0.2 * torch.randn(co, generator=g),
          0.5 + torch.rand(co, generator=g)]
    w16, w8, osc, bias, _ = _prep(nv, w, bn)
    sp = (1,) + shape
    v = (x * A).reshape(N, ci, *sp)
    xh = v.to(torch.float16)
    xl = (v - xh.float()).to(torch.float16)
    vox = int(np.prod(shape))
    xs = torch.cat([_blocked(xh, 8), _blocked(xl, 8)], 1).contiguous().cuda()
    x8 = torch.empty(N * 2 * ci * vox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_make8', nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(x8), 2 * ci * vox, ci, N, 1, shape[0], shape[1], nv.stream())
    hi, lo, y8, sat = _run(nv, xs, x8, w16, w8, osc, bias, N, sp, ci, co, 2, x_ss=2 * ci * vox, nd=2)
    wf, bf = unet_ref.fold_bn(w, *bn)
    xq = ((xh.double() + xl.double()) / A).reshape(N, ci, *shape)
    want = torch.relu(F.conv2d(xq, wf.double(), bf.double(), padding=1)).reshape(N, co, *sp)
    got = (hi.double() + lo.double()) / A
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f'[x2m conv 2-D {ci}->{co} {shape}] max rel err {err:.2e}')
    assert err < 6e-5 and sat == 0, err


@pytest.mark.parametrize('shape,ci,co,N', [((8, 16, 32), 32, 32, 1), ((6, 10, 20), 64, 32, 2), ((4, 8, 16), 96, 64, 1)])
def test_conv3_x2m_random(shape, ci, co, N):
    nv = _nv()
    g = torch.Generator().manual_seed(12)
    x = torch.rand((N, ci) + shape, generator=g) * 2
    w = torch.randn((co, ci, 3, 3, 3), generator=g) * (2.0 / (ci * 27)) ** 0.5
    bn = [0.75 + 0.5 * torch.rand(co, generator=g), 0.1 * torch.randn(co, generator=g), 0.2 * torch.randn(co, generator=g),
          0.5 + torch.rand(co, generator=g)]
    w16, w8, osc, bias, _ = _prep(nv, w, bn)
    v = x * A
    xh = v.to(torch.float16)
    xl = (v - xh.float()).to(torch.float16)
    # the m8 planes through the device kernel that the engine uses behind first conv / pool / transposed conv
    vox = int(np.prod(shape))
    xs = torch.cat([_blocked(xh, 8), _blocked(xl, 8)], 1).contiguous().cuda()
    x8 = torch.empty(N * 2 * ci * vox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_make8', nv.ptr(xs), 2 * ci * vox, ci // 8, nv.ptr(x8), 2 * ci * vox, ci, N, shape[0], shape[1], shape[2], nv.stream())
    lo8 = _m8_unpack(x8.cpu(), N, ci, shape)
    assert torch.equal(lo8, _e4m3(xl.float() * 16.0))
    hi, lo, y8, sat = _run(nv, xs, x8, w16, w8, osc, bias, N, shape, ci, co, 2, x_ss=2 * ci * vox)      # (the lo planes sit behind the hi planes, unread)
    wf, bf = unet_ref.fold_bn(w, *bn)
    xq = (xh.double() + xl.double()) / A
    want = torch.relu(F.conv3d(xq, wf.double(), bf.double(), padding=1))
    got = (hi.double() + lo.double()) / A
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f'[x2m conv {ci}->{co} {shape}] max rel err {err:.2e}')
    assert err < 6e-5, err            # 2^-14; the fp16 mode: 5e-4, a dropped cross term: 2.4e-4
    assert sat == 0


def test_conv3_x2m_without_lo_planes_and_saturation_flag():
    nv = _nv()
    g = torch.Generator().manual_seed(13)
    shape, ci, co, N = (4, 8, 16), 32, 32, 1
    w = torch.randn((co, ci, 3, 3, 3), generator=g).abs() * 0.05
    w16, w8, osc, bias, _ = _prep(nv, w)
    x = torch.rand((N, ci) + shape, generator=g) * 600.0                     # sums far beyond 65504 / act_scale
    v = x * A
    xh = v.to(torch.float16)
    xs = _blocked(xh, 8).cuda()
    x8 = _m8_planes(_e4m3((v - xh.float()) * 16.0)).cuda()
    hi, lo, _, sat = _run(nv, xs, x8, w16, w8, osc, bias, N, shape, ci, co, 0, y_lo=False)
    assert torch.equal(lo, torch.zeros_like(lo))                             # y_lo < 0: no lo planes written
    assert hi.abs().max().item() == 65504.0 and sat == 0x7bff                # the clamp, and its flag


# ---------------------------------------------------------------------------------------------------------------- producers and network
def test_producers_write_the_m8_planes_of_their_hi_and_lo_words():
    """first conv, transposed conv and max-pool in the x2m form against the fp16x2 kernels' hi / lo words (+ iunet_x2m_make8 of them)."""
    from tests.test_gpu_x2 import _prep_conv
    nv = _nv()
    g = torch.Generator().manual_seed(21)
    N, co, shape = 2, 32, (6, 10, 20)
    vox = int(np.prod(shape))
    # ---- first conv
    xu = torch.randint(0, 256, (N, 1) + shape, generator=g, dtype=torch.uint8).cuda()
    w = torch.randn((co, 1, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5
    wpk, osc, b = _prep_conv(nv, w)
    st = nv.ll_array((vox, vox, shape[1] * shape[2], shape[2], 1))
    y = torch.zeros(N * 2 * co * vox, dtype=torch.float16, device='cuda')
    nv.call('iunet_x2_first_conv_fwd', 3, nv.ptr(xu), 2, st, nv.ptr(y), 2 * co * vox, co // 8, nv.ptr(wpk), nv.ptr(osc), nv.ptr(b), A, N, *shape, 1, co, 1, nv.stream())
    want8 = torch.empty(N * 2 * co * vox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_make8', nv.ptr(y), 2 * co * vox, co // 8, nv.ptr(want8), 2 * co * vox, co, N, *shape, nv.stream())
    yh = torch.zeros(N * co * vox, dtype=torch.float16, device='cuda')
    y8 = torch.zeros(N * 2 * co * vox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_first_conv_fwd', 3, nv.ptr(xu), 2, st, nv.ptr(yh), co * vox, -1, nv.ptr(y8), 2 * co * vox, nv.ptr(wpk), nv.ptr(osc), nv.ptr(b), A, N, *shape,
            1, co, 1, None, nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(yh.view(N, co * vox), y.view(N, 2 * co * vox)[:, :co * vox])
    # (the producer rounds the exact fp32 residual, make8 the fp16 lo word of it: equal unless the lo word itself was rounded -- never at these sizes)
    lo_a = _m8_unpack(y8.cpu(), N, co, shape)
    lo_b = _m8_unpack(want8.cpu(), N, co, shape)
    assert (lo_a != lo_b).float().mean().item() < 2e-3
    # ---- max-pool on (hi, m8): the winner's words are copied
    do = tuple(s // 2 for s in shape)
    ovox = int(np.prod(do))
    ph = torch.zeros(N * co * ovox, dtype=torch.float16, device='cuda')
    p8 = torch.zeros(N * 2 * co * ovox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_maxpool_fwd', 3, nv.ptr(yh), co * vox, nv.ptr(y8), 2 * co * vox, nv.ptr(ph), co * ovox, nv.ptr(p8), 2 * co * ovox, co, N, *do, nv.stream())
    torch.cuda.synchronize()
    un = lambda t, sp: t.cpu().reshape(N, co // 8, *sp, 8).permute(0, 1, 5, 2, 3, 4).reshape(N, co, *sp).float()
    hi_full = un(yh, shape)
    val = hi_full + lo_a / 16.0
    want_val = F.max_pool3d(val, 2)
    got_lo = _m8_unpack(p8.cpu(), N, co, do)
    assert torch.equal(un(ph, do) + got_lo / 16.0, want_val)
    # ---- transposed conv
    ci = 64
    xin = torch.rand((N, ci) + do, generator=g) * 2
    wt = torch.randn((ci, co, 2, 2, 2), generator=g) * (1.0 / ci) ** 0.5
    bt = torch.randn(co, generator=g) * 0.1
    wpk, osc, b = _prep_conv(nv, wt, bias=bt, transposed=True)
    from interactive_unet.engine_x2 import EngineX2
    xs = EngineX2(dim=3, mixed=False).to_split(xin).cuda()
    y = torch.zeros(N * 2 * co * ovox * 8, dtype=torch.float16, device='cuda')
    nv.call('iunet_x2_convT_fwd', 3, nv.ptr(xs), 2 * ci * ovox, ci // 8, nv.ptr(y), 2 * co * ovox * 8, co // 8, nv.ptr(wpk), nv.ptr(osc), nv.ptr(b), N, *do, ci, co, nv.stream())
    up = tuple(2 * s for s in do)
    uvox = ovox * 8
    nv.call('iunet_x2m_make8', nv.ptr(y), 2 * co * uvox, co // 8, nv.ptr(want8[:N * 2 * co * uvox]), 2 * co * uvox, co, N, *up, nv.stream())
    yh = torch.zeros(N * co * uvox, dtype=torch.float16, device='cuda')
    y8 = torch.zeros(N * 2 * co * uvox, dtype=torch.uint8, device='cuda')
    nv.call('iunet_x2m_convT_fwd', 3, nv.ptr(xs), 2 * ci * ovox, ci // 8, nv.ptr(yh), co * uvox, -1, nv.ptr(y8), 2 * co * uvox, nv.ptr(wpk), nv.ptr(osc), nv.ptr(b),
            N, *do, ci, co, None, nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(yh.view(N, co * uvox), y.view(N, 2 * co * uvox)[:, :co * uvox])
    lo_a = _m8_unpack(y8.cpu(), N, co, up)
    lo_b = _m8_unpack(want8[:N * 2 * co * uvox].cpu(), N, co, up)
    assert (lo_a != lo_b).float().mean().item() < 2e-3


@pytest.mark.parametrize('shape,cin,ncls,in_dtype', [((16, 32, 48), 1, 3, torch.uint8), ((8, 24, 40), 2, 4, torch.float16),
                                                      ((64, 96), 1, 3, torch.uint8), ((40, 72), 3, 2, torch.float32)])
def test_network_x2m_small_shapes(shape, cin, ncls, in_dtype):
    """The 3-D network in the x2m form against the fp32 oracle (the headline size is in test_gpu_parity.py), the C++-sequenced forward
    bit-identical to the Python-sequenced one, the range flag quiet."""
    from tests.test_gpu_parity import _smooth, _forward, _compare, _assert_fp32_mode, _labels
    from interactive_unet.engine_x2 import EngineX2
    dim = len(shape)
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
    e = EngineX2(dim=dim, cin=cin, ncls=ncls)
    assert e.mixed                                        # the default
    e.load_eval({k: v.cuda() for k, v in p.items()})
    e.use_graph = False
    lg, pr, cl = _forward(e, xd, dim, ncls)
    r = _compare(f'x2m {dim}-D {shape} cin={cin}', lg, pr, cl, ref, _labels(img, ncls))
    _assert_fp32_mode(r)
    assert r['err'] <= 3e-4 * max(1.0, r['scale'])        # ~6e-5 of the logit scale measured; fp16: 1.5e-3
    e.use_graph = True
    e._g_fwd = 1
    lg2, pr2, cl2 = _forward(e, xd, dim, ncls)
    from interactive_unet import net_graph
    assert not net_graph.ENABLED or (e._g is not None and e._g.loaded)
    assert torch.equal(lg, lg2) and torch.equal(pr, pr2) and torch.equal(cl, cl2)
    assert not e.saturated()


def test_network_x2m_c5_geometry_and_range_flag():
    """5 levels, base 64, 4 classes in the x2m form (split-K free: 64 chunk pairs at the widest conv); then the same network with a
    BatchNorm gain that drives activations beyond 65504 / act_scale: finite result, the on-device flag raised."""
    from tests.test_gpu_parity import _smooth, _forward, _compare, _assert_fp32_mode, _labels
    from interactive_unet.engine_x2 import EngineX2
    dim, shape, ncls = 3, (16, 16, 32), 4
    p = unet_ref.init_params(dim=dim, levels=5, base=64, ncls=ncls, seed=4, randomize_bn=True)
    img = np.stack([_smooth(shape, 3 + i) for i in range(1)])[:, None]
    x = torch.tensor(img)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=dim, levels=5)
    e = EngineX2(dim=dim, levels=5, base=64, ncls=ncls)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    r = _compare('x2m 3-D 5 levels base 64', *_forward(e, x.cuda(), dim, ncls), ref, _labels(img, ncls))
    _assert_fp32_mode(r)
    assert not e.saturated()
    p2 = dict(p)
    p2['enc0.bn2.weight'] = p['enc0.bn2.weight'] * 4000.0
    e2 = EngineX2(dim=dim, levels=5, base=64, ncls=ncls)
    e2.load_eval({k: v.cuda() for k, v in p2.items()})
    lg, _, _ = _forward(e2, x.cuda(), dim, ncls)
    assert torch.isfinite(lg).all() and e2.saturated()


@pytest.mark.parametrize('dim,shape,ncls', [(3, (16, 32, 48), 2), (3, (8, 24, 40), 3), (2, (64, 96), 2), (2, (40, 72), 3)])
def test_head_in_the_last_conv_epilogue_is_conv_plus_head_bit_for_bit(dim, shape, ncls, monkeypatch):
    """iunet_x2m_conv_head_fwd (the last stage conv with the 1x1 head + softmax + class map in its epilogue, the last activation never
    written) against iunet_x2m_conv_fwd into hi + lo planes followed by iunet_x2_head_fwd: logits, probabilities (accumulating into a
    channels-last buffer with a divisor, the 2.5-D / block-store contract) and class map must be the same bits."""
    nv = _nv()
    assert nv.lib().iunet_x2m_head_fusable(ncls, 32) == 1 and nv.lib().iunet_x2m_head_fusable(5, 32) == 0
    g = torch.Generator().manual_seed(41)
    N, ci, co = 2, 32, 32
    sp = shape if dim == 3 else (1,) + shape
    vox = int(np.prod(shape))
    x = torch.rand((N, ci) + sp, generator=g) * 2
    w = torch.randn((co, ci) + (3,) * dim, generator=g) * (2.0 / (ci * 3 ** dim)) ** 0.5
    bn = [0.75 + 0.5 * torch.rand(co, generator=g), 0.1 * torch.randn(co, generator=g), 0.2 * torch.randn(co, generator=g), 0.5 + torch.rand(co, generator=g)]
    w16, w8, osc, bias, _ = _prep(nv, w, bn)
    v = x * A
    xh = v.to(torch.float16)
    xs = _blocked(xh, 8).cuda()
    x8 = _m8_planes(_e4m3((v - xh.float()) * 16.0)).cuda()
    hw = (torch.randn((ncls, co), generator=g) * 0.3).cuda()
    hb = (torch.randn(ncls, generator=g) * 0.1).cuda()
    # unfused: conv -> hi + lo planes -> head
    y = torch.zeros(N * 2 * co * vox, dtype=torch.float16, device='cuda')
    nv.call('iunet_x2m_conv_fwd', dim, nv.ptr(xs), ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), 2 * co * vox, co // 8, None, 0, nv.ptr(w16), nv.ptr(w8),
            nv.ptr(osc), nv.ptr(bias), N, *sp, ci, co, 2, None, nv.stream())
    outs = []
    for fused in (False, True):
        lg = torch.zeros((N, ncls) + sp, device='cuda')
        pr = torch.full((N,) + sp + (ncls,), 0.125, device='cuda')                 # channels-last, accumulated into
        cl = torch.zeros((N, vox), dtype=torch.uint8, device='cuda')
        st_l = nv.ll_array((ncls * vox, vox, sp[1] * sp[2], sp[2], 1))
        st_p = nv.ll_array((ncls * vox, 1, sp[1] * sp[2] * ncls, sp[2] * ncls, ncls))
        for out, st in ((('lg',), st_l), (('pr',), st_p)):
            args = (nv.ptr(lg) if out[0] == 'lg' else None, nv.ptr(pr) if out[0] == 'pr' else None, nv.ptr(cl), st, 3.0 if out[0] == 'pr' else 1.0,
                    1 if out[0] == 'pr' else 0)
            if fused:
                nv.call('iunet_x2m_conv_head_fwd', dim, nv.ptr(xs), ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias),
                        nv.ptr(hw), nv.ptr(hb), A, ncls, *args, N, *sp, ci, None, nv.stream())
            else:
                nv.call('iunet_x2_head_fwd', nv.ptr(y), 2 * co * vox, co // 8, co, nv.ptr(hw), nv.ptr(hb), A, ncls, *args, N, *sp, nv.stream())
        torch.cuda.synchronize()
        outs.append((lg.cpu(), pr.cpu(), cl.cpu()))
    (lg0, pr0, cl0), (lg1, pr1, cl1) = outs
    assert torch.equal(lg0, lg1) and torch.equal(pr0, pr1) and torch.equal(cl0, cl1)
    assert lg0.abs().max() > 0.1 and (pr0 != 0.125 / 3.0).any()


@pytest.mark.parametrize('nd,shape,ci,co,N', [(3, (8, 16, 32), 32, 32, 2), (3, (6, 10, 20), 64, 32, 1), (3, (16, 32, 48), 32, 64, 3), (3, (4, 8, 16), 96, 64, 1),
                                            (2, (1, 32, 64), 32, 32, 2), (2, (1, 24, 40), 64, 64, 1), (2, (1, 70, 132), 32, 32, 3)])
def test_max_pool_in_the_conv_epilogue_is_conv_plus_pool_bit_for_bit(nd, shape, ci, co, N):
    """iunet_x2m_conv_pool_fwd: the encoder stages' second conv with the 2^d max-pool riding in its epilogue (2-D: in the consumer waves'
    registers; 3-D: x, y in registers, the z pair through LDS by the loader waves; big and small tiles, ragged grids, several samples and
    Cout tiles) writes the conv's own output unchanged and the words iunet_x2m_maxpool_fwd makes of it.  The data hold many exact ties
    (ReLU zeros, repeated values): the winner is defined by a total order on the words, not by position."""
    nv = _nv()
    g = torch.Generator().manual_seed(31)
    sp = shape[3 - nd:]
    x = torch.rand((N, ci) + shape, generator=g) * 2
    w = torch.randn((co, ci) + (3,) * nd, generator=g) * (2.0 / (ci * 3 ** nd)) ** 0.5
    bn = [0.75 + 0.5 * torch.rand(co, generator=g), 0.1 * torch.randn(co, generator=g) - 0.3, 0.2 * torch.randn(co, generator=g),
          0.5 + torch.rand(co, generator=g)]
    w16, w8, osc, bias, _ = _prep(nv, w, bn)
    v = x * A
    xh = v.to(torch.float16)
    vox = int(np.prod(shape))
    xs = _blocked(xh, 8).cuda()
    x8 = _m8_planes(_e4m3((v - xh.float()) * 16.0)).cuda()
    po = tuple(s // 2 if i >= 3 - nd else 1 for i, s in enumerate(shape))
    pvox = int(np.prod(po))
    sat = torch.zeros(1, dtype=torch.int32, device='cuda')

    def conv(pool):
        y = torch.zeros(N * co * vox, dtype=torch.float16, device='cuda')
        y8 = torch.zeros(N * 2 * co * vox, dtype=torch.uint8, device='cuda')
        py = torch.full((N * co * pvox,), 7.0, dtype=torch.float16, device='cuda')
        py8 = torch.full((N * 2 * co * pvox,), 9, dtype=torch.uint8, device='cuda')
        if pool:
            nv.call('iunet_x2m_conv_pool_fwd', nd, nv.ptr(xs), ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), co * vox, -1, nv.ptr(y8), 2 * co * vox,
                    nv.ptr(py), co * pvox, nv.ptr(py8), 2 * co * pvox, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias),
                    N, shape[0], shape[1], shape[2], ci, co, 2, nv.ptr(sat), nv.stream())
        else:
            nv.call('iunet_x2m_conv_fwd', nd, nv.ptr(xs), ci * vox, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), co * vox, -1, nv.ptr(y8), 2 * co * vox,
                    nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, shape[0], shape[1], shape[2], ci, co, 2, nv.ptr(sat), nv.stream())
            nv.call('iunet_x2m_maxpool_fwd', nd, nv.ptr(y), co * vox, nv.ptr(y8), 2 * co * vox, nv.ptr(py), co * pvox, nv.ptr(py8), 2 * co * pvox,
                    co, N, po[0], po[1], po[2], nv.stream())
        torch.cuda.synchronize()
        return y, y8, py, py8

    a, b = conv(False), conv(True)
    for name, u, f in zip(('hi', 'm8', 'pooled hi', 'pooled m8'), a, b):
        assert torch.equal(u, f), f'{name}: {int((u != f).sum())} of {u.numel()} words differ'
    # and the pooled words are the max-pool of the values a 3x3x3 consumer reads: hi + lo8 / 16
    lo8 = _m8_unpack(a[1].cpu(), N, co, shape)
    val = a[0].cpu().float().reshape(N, co // 8, *shape, 8).permute(0, 1, 5, 2, 3, 4).reshape(N, co, *shape) + lo8 / 16.0
    plo8 = _m8_unpack(b[3].cpu(), N, co, po)
    ph = b[2].cpu().float().reshape(N, co // 8, *po, 8).permute(0, 1, 5, 2, 3, 4).reshape(N, co, *po)
    want = F.max_pool3d(val, (1, 2, 2) if nd == 2 else 2)
    assert torch.equal(ph + plo8 / 16.0, want)
    assert (want == 0).float().mean() > 0.02 and sat.item() == 0          # (the ReLU zeros are there: ties were exercised)


@pytest.mark.parametrize('shape,N,in_dtype,pool', [((64, 96), 2, torch.uint8, True), ((70, 132), 3, torch.float32, True), ((18, 34), 1, torch.float16, False),
                                                   ((256, 256), 2, torch.uint8, True)])
def test_first_encoder_stage_in_one_launch_is_first_conv_plus_conv_bit_for_bit(shape, N, in_dtype, pool):
    """iunet_x2m_first_stage_fwd (2-D): the second conv of the first encoder stage whose loader waves compute the first conv (1 -> 32 channels)
    on the way in -- the tensor between the two convs never exists -- against iunet_x2m_first_conv_fwd + iunet_x2m_conv_fwd (+ pool): the skip
    tensor and the pooled tensor, every word.  Strided input views of three dtypes, ragged tile grids (halo rows and columns outside the image
    are the second conv's zero padding, not conv values), several tiles per workgroup."""
    from tests.test_gpu_x2 import _prep_conv
    nv = _nv()
    g = torch.Generator().manual_seed(51)
    H, W = shape
    vox, c = H * W, 32
    if in_dtype == torch.uint8:
        base = torch.randint(0, 256, (N, H + 3, W + 5), generator=g, dtype=torch.uint8).cuda()
    else:
        base = torch.rand((N, H + 3, W + 5), generator=g).to(in_dtype).cuda()
    x = base[:, 1:H + 1, 2:W + 2]                                   # a strided view: the kernels read the caller's tensor as it lies
    st = nv.ll_array((x.stride(0), x.stride(0), x.stride(0), x.stride(1), x.stride(2)))
    w1 = torch.randn((c, 1, 3, 3), generator=g) * (2.0 / 9) ** 0.5
    bn1 = [0.75 + 0.5 * torch.rand(c, generator=g), 0.1 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g), 0.5 + torch.rand(c, generator=g)]
    fw, fosc, fb = _prep_conv(nv, w1, bn1)
    w2 = torch.randn((c, c, 3, 3), generator=g) * (2.0 / (c * 9)) ** 0.5
    bn2 = [0.75 + 0.5 * torch.rand(c, generator=g), 0.1 * torch.randn(c, generator=g) - 0.2, 0.2 * torch.randn(c, generator=g), 0.5 + torch.rand(c, generator=g)]
    w16, w8, osc, bias, _ = _prep(nv, w2, bn2)
    f = nv.lib().iunet_x2m_first_stage_fusable
    assert f(3, 1, 32, 8, 512, 512) == 0 and f(2, 2, 32, 8, 512, 512) == 0
    if 'IUNET_X2M_FIRST' not in os.environ:          # the library's own policy: batches of >= 2 048 tiles whose pool does not ride in the second conv
        assert f(2, 1, 32, 8, 512, 512) == (0 if nv.lib().iunet_x2m_pool_fusable(2, 32) else 1) and f(2, 1, 32, 1, 128, 128) == 0
    pv = (H // 2) * (W // 2)
    code = nv.IN_DTYPE_CODE[x.dtype]

    def run(fused):
        y = torch.zeros(N * c * vox, dtype=torch.float16, device='cuda')
        y8 = torch.zeros(N * 2 * c * vox, dtype=torch.uint8, device='cuda')
        py = torch.full((N * c * pv,), 7.0, dtype=torch.float16, device='cuda')
        py8 = torch.full((N * 2 * c * pv,), 9, dtype=torch.uint8, device='cuda')
        sat = torch.zeros(1, dtype=torch.int32, device='cuda')
        pargs = (nv.ptr(py), c * pv, nv.ptr(py8), 2 * c * pv) if pool else (None, 0, None, 0)
        if fused:
            nv.call('iunet_x2m_first_stage_fwd', nv.ptr(x), code, st, nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb), A, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox,
                    *pargs, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, H, W, nv.ptr(sat), nv.stream())
        else:
            a = torch.zeros(N * c * vox, dtype=torch.float16, device='cuda')
            a8 = torch.zeros(N * 2 * c * vox, dtype=torch.uint8, device='cuda')
            nv.call('iunet_x2m_first_conv_fwd', 2, nv.ptr(x), code, st, nv.ptr(a), c * vox, -1, nv.ptr(a8), 2 * c * vox, nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb), A,
                    N, 1, H, W, 1, c, 1, nv.ptr(sat), nv.stream())
            if pool:
                nv.call('iunet_x2m_conv_pool_fwd', 2, nv.ptr(a), c * vox, nv.ptr(a8), 2 * c * vox, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox, *pargs,
                        nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, 1, H, W, c, c, 2, nv.ptr(sat), nv.stream())
            else:
                nv.call('iunet_x2m_conv_fwd', 2, nv.ptr(a), c * vox, nv.ptr(a8), 2 * c * vox, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox,
                        nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, 1, H, W, c, c, 2, nv.ptr(sat), nv.stream())
        torch.cuda.synchronize()
        return y, y8, py, py8, sat

    a, b = run(False), run(True)
    for name, u, f in zip(('hi', 'm8', 'pooled hi', 'pooled m8', 'range flag'), a, b):
        assert torch.equal(u, f), f'{name}: {int((u != f).sum())} of {u.numel()} words differ'
    assert a[0].float().abs().max().item() > 0 and (a[0] == 0).float().mean().item() < 0.9


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_FIRST_STAGE_CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/interactive-unet_amd')
from interactive_unet.engine_x2 import EngineX2
from interactive_unet import _native as nv
from oracle import unet_ref
assert nv.lib().iunet_x2m_first_stage_fusable(2, 1, 32, 8, 512, 512) == 1
p = unet_ref.init_params(dim=2, seed=3, randomize_bn=True)
e = EngineX2(dim=2)
e.load_eval({k: v.cuda() for k, v in p.items()})
x = torch.randint(0, 256, (8, 1, 512, 512), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).cuda()
vox = 512 * 512
outs = []
for rep in range(2):                                   # second call: the C++-sequenced forward (net_graph)
    lg = torch.empty((8, 2, 512, 512), device='cuda')
    e.infer(x, (vox, vox, vox, 512, 1), 8, 1, 512, 512, logits=lg)
    outs.append(lg.cpu())
assert not e.saturated()
torch.save(outs, sys.argv[2])
'''


def test_network_2d_with_the_first_stage_in_one_launch_equals_the_library_policy(tmp_path):
    """The library's own policy runs encoder stage 0 as first conv + pooled conv (iunet_x2m_first_stage_fusable: at 3 bytes per element
    that is the faster sequence wherever the pool rides in the conv); IUNET_X2M_FIRST=2 -- read once per process, hence the child -- runs the
    stage as ONE launch.  Same logits, bit for bit, through the Python-sequenced forward and the C++-sequenced one; a slice alone equals its
    row of the batch."""
    import subprocess, sys
    from interactive_unet.engine_x2 import EngineX2
    nv = _nv()
    p = unet_ref.init_params(dim=2, seed=3, randomize_bn=True)
    e = EngineX2(dim=2)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (8, 1, 512, 512), generator=g, dtype=torch.uint8).cuda()
    vox = 512 * 512
    outs = []
    for rep in range(2):                                   # second call: the C++-sequenced forward (net_graph)
        lg = torch.empty((8, 2, 512, 512), device='cuda')
        e.infer(x, (vox, vox, vox, 512, 1), 8, 1, 512, 512, logits=lg)
        outs.append(lg)
    one = torch.empty((1, 2, 512, 512), device='cuda')
    for i in (0, 5):
        e.infer(x[i:i + 1], (vox, vox, vox, 512, 1), 1, 1, 512, 512, logits=one)
        torch.cuda.synchronize()
        assert torch.equal(one[0], outs[0][i]) and torch.equal(one[0], outs[1][i])
    assert not e.saturated()
    if os.environ.get('IUNET_X2M_FIRST') == '2':
        return                                             # this process already ran the one-launch stage
    out = tmp_path / 'first_stage_logits.pt'
    env = dict(os.environ, IUNET_X2M_FIRST='2')
    r = subprocess.run([sys.executable, '-c', _FIRST_STAGE_CHILD, ROOT, str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    child = torch.load(out)
    for k in range(2):
        assert torch.equal(child[k], outs[k].cpu()), f'forward {k}: {int((child[k] != outs[k].cpu()).sum())} logits differ'


@pytest.mark.parametrize('dim,mixed', [(3, True), (2, True), (3, False), (2, False)])
def test_batched_operator_preparation_equals_the_per_layer_calls(dim, mixed, monkeypatch):
    """EngineX2.load_eval re-prepares every operator of the network in three launches over descriptor tables (iunet_x2_prep_batch,
    iunet_x2m_prep_batch, iunet_pack_batch); IUNET_X2_PREP_PER_LAYER=1 makes the two calls per operator they replace.  Same packed operators,
    scales and biases, bit for bit -- and the tables follow the parameters in place (an optimiser step moves values, not addresses)."""
    from interactive_unet.engine_x2 import EngineX2
    p = {k: v.cuda() for k, v in unet_ref.init_params(dim=dim, seed=11, randomize_bn=True).items()}
    e = EngineX2(dim=dim, mixed=mixed)

    def snapshot():
        torch.cuda.synchronize()
        return {k: [t.clone() for t in v] for k, v in e.packed.items() if k != 'head'}
    monkeypatch.delenv('IUNET_X2_PREP_PER_LAYER', raising=False)
    e.load_eval(p)
    assert len(e._eval_tables) == (3 if mixed else 2)
    a = snapshot()
    for k, v in e.packed.items():                   # the second pass must rewrite everything
        if k != 'head':
            for t in v:
                t.zero_() if t.dtype == torch.uint8 else t.fill_(3)          # (the K = 128 operator's padding bytes stay zero)
    monkeypatch.setenv('IUNET_X2_PREP_PER_LAYER', '1')
    e.load_eval(p)
    b = snapshot()
    assert a.keys() == b.keys() and len(a) == 4 * (e.levels - 1) + 2 + (e.levels - 1)
    for k in a:
        for i, (u, v) in enumerate(zip(a[k], b[k])):
            assert torch.equal(u, v), f'{k}[{i}]: {int((u != v).sum())} of {u.numel()} words differ'
    # new values at the same addresses: the cached tables see them
    with torch.no_grad():
        for k, v in p.items():
            if k.endswith('weight') and v.dim() > 1:
                v.mul_(1.25).add_(0.01)
    monkeypatch.delenv('IUNET_X2_PREP_PER_LAYER')
    tabs = e._eval_tables
    e.load_eval(p)
    assert e._eval_tables is tabs
    c = snapshot()
    monkeypatch.setenv('IUNET_X2_PREP_PER_LAYER', '1')
    for k, v in e.packed.items():
        if k != 'head':
            for t in v:
                t.zero_() if t.dtype == torch.uint8 else t.fill_(5)
    e.load_eval(p)
    d = snapshot()
    assert any(not torch.equal(a[k][0], c[k][0]) for k in a)
    for k in c:
        for i, (u, v) in enumerate(zip(c[k], d[k])):
            assert torch.equal(u, v), f'after the update, {k}[{i}]: {int((u != v).sum())} of {u.numel()} words differ'
