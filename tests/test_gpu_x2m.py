"""fp16x2 with the cross terms on the fp8 matrix cores ("x2m": csrc/conv3_x2m.hip, EngineX2(mixed=True)): the 3x3x3 stage conv of
the split-precision forward as x_hi w_hi on v_mfma_f32_16x16x32_f16 + [x_lo8 | x_hi8] [w_hi8 | w_lo8] on v_mfma_f32_16x16x128_f8f6f4.

Kernel level: (a) data on which every operand of BOTH steps is a small integer -- hi planes, hand-made m8 planes, operator entries of the
form 16 a + b / 256 so that w_hi8 = a and w_lo8 = b exactly -- must give the bit-exact conv (lane maps of the two instructions, operator
orders, the pairing of the powers of two, halos, ragged tiles); (b) on random fp32 data the result is within ~2^-14 of the exact conv
of the split input (the fp16 mode leaves 2^-11; a dropped cross term 2^-12).  Network level: logits against the fp32 oracle.
"""
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


def _m8_planes(lo8_vals, hi8_vals):
    """[N, C, D, H, W] e4m3-representable values -> m8 byte tensor [N][2 C / 16][D][H][W][16]: plane 2c = lo8 of chunk c, 2c + 1 = hi8."""
    N, C = lo8_vals.shape[:2]
    lo = _blocked(lo8_vals, 16)
    hi = _blocked(hi8_vals, 16)
    both = torch.stack([lo, hi], 2).reshape(N, 2 * C // 16, *lo.shape[2:])
    return both.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


def _m8_unpack(b, N, C, sp):
    t = b.view(torch.float8_e4m3fn).float().reshape(N, C // 16, 2, *sp, 16)
    un = lambda u: u.permute(0, 1, 5, 2, 3, 4).reshape(N, C, *sp)
    return un(t[:, :, 0]), un(t[:, :, 1])


def _prep(nv, w, bn=None, act_out=A):
    co, ci = w.shape[:2]
    dev = 'cuda'
    w = w.to(dev, torch.float32).contiguous()
    whi = torch.empty(co * ci * 27, device=dev)
    w8 = torch.zeros(nv.lib().iunet_x2m_w8_bytes(co, ci), dtype=torch.uint8, device=dev)
    osc, b = torch.empty(co, device=dev), torch.empty(co, device=dev)
    bnp = [None] * 4 if bn is None else [t.to(dev, torch.float32).contiguous() for t in bn]
    nv.call('iunet_x2m_prep', nv.ptr(w), nv.ptr(whi), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), *[nv.ptr(t) for t in bnp], 1e-5, A, act_out, co, ci, nv.stream())
    w16 = torch.empty(nv.pack_conv3_elems(co, ci, 27, 2), dtype=torch.float16, device=dev)
    nv.call('iunet_pack_conv3', 0, nv.ptr(whi), None, nv.ptr(w16), co, ci, 27, 2, nv.stream())
    torch.cuda.synchronize()
    return w16, w8, osc, b, whi.reshape(co, ci, 3, 3, 3).cpu()


def _run(nv, xhi, x8, w16, w8, osc, bias, N, shape, ci, co, epi, y_lo=True, x_ss=None):
    vox = int(np.prod(shape))
    x_ss = ci * vox if x_ss is None else x_ss
    y = torch.zeros(N * 2 * co * vox, dtype=torch.float16, device='cuda')
    y8 = torch.zeros(N * 2 * co * vox, dtype=torch.uint8, device='cuda')
    sat = torch.zeros(1, dtype=torch.int32, device='cuda')
    nv.call('iunet_x2m_conv3_fwd', nv.ptr(xhi), x_ss, nv.ptr(x8), 2 * ci * vox, nv.ptr(y), 2 * co * vox, co // 8 if y_lo else -1,
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
    # operator entries 16 a + b / 256: |a| in {32, 36, .., 60} (hi = 16 a exactly, a is an e4m3 value), |b| <= 15 (w_lo8 = b)
    a = (torch.randint(8, 16, (co, ci, 3, 3, 3), generator=g) * 4).float() * (torch.randint(0, 2, (co, ci, 3, 3, 3), generator=g) * 2 - 1).float()
    b = torch.randint(-15, 16, (co, ci, 3, 3, 3), generator=g).float()
    w = 16.0 * a + b / 256.0
    w16, w8, osc, bias, whi = _prep(nv, w, act_out=1.0)                      # accumulator / 64: the sums stay below the fp16 range
    assert torch.equal(whi, 16.0 * a)                                        # the row scale is 1 (max |w| in [2^9, 2^10))
    assert torch.equal(osc.cpu(), torch.full((co,), 1.0 / 64)) and torch.equal(bias.cpu(), torch.zeros(co))
    X = torch.randint(-3, 4, (N, ci) + shape, generator=g).float()            # hi planes (the main term's operand)
    L8 = torch.randint(-4, 5, (N, ci) + shape, generator=g).float()           # lo8 plane values
    H8 = torch.randint(-4, 5, (N, ci) + shape, generator=g).float()           # hi8 plane values (independent of X on purpose)
    xhi = _blocked(X, 8).to(torch.float16).cuda()
    x8 = _m8_planes(L8, H8).cuda()
    hi, lo, y8, sat = _run(nv, xhi, x8, w16, w8, osc, bias, N, shape, ci, co, 0)
    want = (F.conv3d(X.double(), (16.0 * a).double(), padding=1) + F.conv3d(L8.double(), a.double(), padding=1)
            + F.conv3d(H8.double(), b.double(), padding=1))
    assert want.abs().max() < 2 ** 22                                        # 22 bits: hi + lo hold acc / 64 exactly
    got = (hi.double() + lo.double()) * 64.0
    assert torch.equal(got, want), (got - want).abs().max()
    # the m8 planes of the output: e4m3 of (v - hi) * 16 and of hi / 256
    v = (want / 64.0).float()
    h = v.to(torch.float16).float()
    lo8, hi8 = _m8_unpack(y8, N, co, shape)
    assert torch.equal(hi, h)
    assert torch.equal(lo8, _e4m3((v - h) * 16.0)) and torch.equal(hi8, _e4m3(h / 256.0))
    assert sat == 0


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
    lo8, hi8 = _m8_unpack(x8.cpu(), N, ci, shape)
    assert torch.equal(lo8, _e4m3(xl.float() * 16.0)) and torch.equal(hi8, _e4m3(xh.float() / 256.0))
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
    x8 = _m8_planes(_e4m3((v - xh.float()) * 16.0), _e4m3(xh.float() / 256.0)).cuda()
    hi, lo, _, sat = _run(nv, xs, x8, w16, w8, osc, bias, N, shape, ci, co, 0, y_lo=False)
    assert torch.equal(lo, torch.zeros_like(lo))                             # y_lo < 0: no lo planes written
    assert hi.abs().max().item() == 65504.0 and sat == 0x7bff                # the clamp, and its flag
