"""BASELINE config C5 on the fp8 matrix cores (csrc/conv3_f8.hip): operator stored as OCP e4m3 bytes + per-output-channel
power-of-two scales, 16-bit activations rounded to e4m3 on their way into LDS, v_mfma_f32_16x16x32_fp8_fp8.

* packing: the bytes, read back as torch.float8_e4m3fn, times the scales are EXACTLY the oracle's quantisation of the
  (BatchNorm-folded) fp32 weights (oracle/unet_ref.quantize_e4m3, itself pinned against torch's e4m3 cast in the CPU suite);
* kernel: small-integer data, where every e4m3 product and fp32 sum is exact -> bit equality with torch's fp32 conv
  (fragment map of the fp8 MFMA, taps, halos, resident / streamed weights, the half-size tile);
* rounding of the activations: values that are NOT e4m3-representable -> equality with the conv of the oracle's rounded
  input (round to nearest even, saturation at 448);
* network: C5's architecture against the CPU emulation with the same quantised operands, tolerance stated there.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_ref
from tests.test_gpu_kernels import blocked, unblocked, nv      # noqa: F401  (fixture + layout helpers)

DT = {'f16': torch.float16, 'bf16': torch.bfloat16}


def run_conv_f8(nv, x, w, dtype, nd, bn=None, epi=0, bias=None, splitk=True):
    """x [N,Cin,*sp] fp32 cpu (values exact in `dtype`), w [Cout,Cin,k..] -> (y fp32 cpu, bytes uint8, scale fp32, bias)."""
    dev = 'cuda'
    N, Cin = x.shape[:2]
    Cout = w.shape[0]
    sp = tuple(x.shape[2:])
    D, H, W = sp if nd == 3 else (1,) + sp
    taps, vox = 3 ** nd, D * H * W
    wd = w.contiguous().to(dev)
    dst = torch.zeros(nv.lib().iunet_f8_pack_conv3_bytes(Cout, Cin, taps), dtype=torch.uint8, device=dev)
    sc = torch.empty(Cout, device=dev)
    b_out = torch.empty(Cout, device=dev)
    bnp = [None] * 4 if bn is None else [nv.ptr(t.to(dev)) for t in bn]
    keep = None if bn is None else [t.to(dev) for t in bn]
    if bn is not None:
        bnp = [nv.ptr(t) for t in keep]
    nv.call('iunet_f8_pack_conv3', nv.ptr(wd), bnp[0], bnp[1], bnp[2], bnp[3], 1e-5, nv.ptr(dst), nv.ptr(sc),
            nv.ptr(b_out) if bn is not None else None, Cout, Cin, taps, nv.stream())
    bd = b_out if bn is not None else (None if bias is None else bias.to(dev))
    xb = blocked(x, dtype).to(dev)
    y = torch.full((N * Cout * vox,), float('nan'), dtype=dtype, device=dev)
    need = nv.lib().iunet_conv3_f8_workspace_elems(nd, N, D, H, W, Cin, Cout) if splitk else 0
    ws = torch.full((need,), float('nan'), device=dev) if need else None
    nv.call('iunet_conv3_f8_fwd', nv.DTYPE_CODE[dtype], nd, nv.ptr(xb), Cin * vox, nv.ptr(y), Cout * vox, nv.ptr(dst), nv.ptr(sc),
            nv.ptr(bd), N, D, H, W, Cin, Cout, epi, nv.ptr(ws), nv.stream())
    torch.cuda.synchronize()
    return unblocked(y.float().cpu(), N, Cout, sp), dst.cpu(), sc.cpu(), (None if bd is None else bd.cpu())


F8K_COL = ((0, 3, 1, 4), (2, 5, 6, 7))       # filter column (dz * 3 + dx) of K = 128 group g, lane group q; column 8: the K = 32 part


def unpack_k128(bytes_u8, cout, cin):
    """Inverse of the K128 order (include/iunet.h: iunet_f8_pack_order; csrc/conv3_f8k.hip) -> float32 [Cout][Cin][27]."""
    vals = bytes_u8.view(torch.float8_e4m3fn).float().numpy()
    nblk = (cout // 32) * (cin // 32)
    v = vals[:nblk * 27648].reshape(cout // 32, cin // 32, 27648)
    big = v[:, :, :24576].reshape(cout // 32, cin // 32, 2, 3, 2, 2, 64, 16)      # [g][dy][m][half e][lane][16 channels]
    small = v[:, :, 24576:].reshape(cout // 32, cin // 32, 3, 2, 64, 8)          # [dy][m][lane][8 channels]
    out = np.zeros((cout, cin, 27), np.float32)
    for lane in range(64):
        row, q = lane & 15, lane >> 4
        for m in range(2):
            co_in = 8 * (row >> 2) + 4 * m + (row & 3)
            for dy in range(3):
                for g in range(2):
                    col = F8K_COL[g][q]
                    tap = ((col // 3) * 3 + dy) * 3 + col % 3
                    for e in range(2):
                        for j in range(16):
                            out[co_in::32, 16 * e + j::32, tap] = big[:, :, g, dy, m, e, lane, j]
                tap = (2 * 3 + dy) * 3 + 2                                       # column 8 = (dz 2, dx 2)
                for j in range(8):
                    out[co_in::32, 8 * q + j::32, tap] = small[:, :, dy, m, lane, j]
    return out


def unpack_k16(bytes_u8, cout, cin, taps):
    """Inverse of the layer's operator order -> float32 [Cout][Cin][taps] (e4m3 values): the K16 order
    [cob32][chunk16][column pair][dy][2][64][8], or the K128 order where iunet_f8_pack_order says so."""
    from interactive_unet import _native
    if _native.lib().iunet_f8_pack_order(taps, cin):
        return unpack_k128(bytes_u8, cout, cin)
    vals = bytes_u8.view(torch.float8_e4m3fn).float().numpy()
    ncol = taps // 3
    ncmb, nchunk = (ncol + 1) // 2, cin // 16
    v = vals.reshape(cout // 32, nchunk, ncmb, 3, 2, 64, 8)
    out = np.zeros((cout, cin, taps), np.float32)
    for lane in range(64):
        row, qq = lane & 15, lane >> 4
        for m in range(2):
            co_in = 8 * (row >> 2) + 4 * m + (row & 3)
            for c in range(ncmb):
                col = 2 * c + (qq >> 1)
                if col >= ncol:
                    assert not v[:, :, c, :, m, lane, :].any()        # the missing column of the last pair is zero
                    continue
                for dy in range(3):
                    tap = ((col // 3) * 3 + dy) * 3 + col % 3
                    for j in range(8):
                        out[co_in::32, 8 * (qq & 1) + j::16, tap] = v[:, :, c, dy, m, lane, j]
    return out


@pytest.mark.parametrize('nd,cout,cin', [(3, 32, 32), (3, 64, 256), (2, 64, 64)])
def test_packed_bytes_are_the_oracle_quantisation(nv, nd, cout, cin):
    g = torch.Generator().manual_seed(3)
    taps = 3 ** nd
    w = torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.05
    w[0] *= 40.0                                                     # channels with very different ranges -> different scales
    w[1] *= 1e-3
    bn = [0.75 + 0.5 * torch.rand(cout, generator=g), 0.1 * torch.randn(cout, generator=g),
          0.2 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g)]
    x = torch.zeros((1, cin) + ((4, 8, 16) if nd == 3 else (16, 32)))
    _, bytes_u8, sc, bias = run_conv_f8(nv, x, w, torch.bfloat16, nd, bn=bn)
    wf, bf = unet_ref.fold_bn_exact(w, *bn)
    want = unet_ref.quantize_e4m3(wf).numpy().reshape(cout, cin, taps)          # scale x e4m3 value, fp32
    got = unpack_k16(bytes_u8, cout, cin, taps) * sc.numpy()[:, None, None]
    assert np.array_equal(got, want)
    assert np.array_equal(bias.numpy(), bf.numpy())
    assert set(np.unique(np.log2(sc.numpy()) % 1)) == {0.0}         # powers of two


@pytest.mark.parametrize('nd,cout,cin,fold', [(3, 32, 32, True), (3, 128, 64, True), (2, 64, 96, False), (3, 64, 512, True)])
def test_pack_table_kind5_equals_the_per_layer_entry(nv, nd, cout, cin, fold):
    """The engine packs its e4m3 operators through the descriptor table (iunet_pack_batch, kind 5, LDS-staged): bytes, scales and
    folded bias must equal iunet_f8_pack_conv3's, which the test above pins to the oracle.  The weight tensor sits at an odd
    float offset of its storage once (the scalar load path of the staging)."""
    g = torch.Generator().manual_seed(11)
    taps, dev = 3 ** nd, 'cuda'
    for off in (0, 3):
        store = torch.zeros(off + cout * cin * taps, device=dev)
        w = store[off:].view((cout, cin) + (3,) * nd)
        w.copy_(torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.05)
        w[0] *= 30.0
        bn = [t.to(dev) for t in (0.75 + 0.5 * torch.rand(cout, generator=g), 0.1 * torch.randn(cout, generator=g),
                                  0.2 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g))] if fold else None
        nb = nv.lib().iunet_f8_pack_conv3_bytes(cout, cin, taps)
        ref, got = (torch.full((nb,), 0xAA, dtype=torch.uint8, device=dev) for _ in range(2))
        sc_ref, sc_got = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        b_ref, b_got = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        bp = [None] * 4 if bn is None else [nv.ptr(t) for t in bn]
        nv.call('iunet_f8_pack_conv3', nv.ptr(w), bp[0], bp[1], bp[2], bp[3], 1e-5, nv.ptr(ref), nv.ptr(sc_ref),
                nv.ptr(b_ref) if fold else None, cout, cin, taps, nv.stream())
        desc = nv.make_desc(w, got, cout, cin, taps, 5, torch.bfloat16, bn=bn, bias_out=b_got if fold else None, eps=1e-5, qscale=sc_got)
        nv.PackTable([desc], dev, sources=[store, got, sc_got, b_got] + (bn or [])).run()
        torch.cuda.synchronize()
        assert torch.equal(sc_got, sc_ref)
        assert torch.equal(got, ref), (off, int((got != ref).sum()))
        if fold:
            assert torch.equal(b_got, b_ref)


def blocked_q(t):
    """[N, C, *spatial] float tensor of e4m3-representable values -> flat uint8 e4m3 planes [N][C / 16][*spatial][16] (format 1)."""
    N, C = t.shape[:2]
    sp = t.shape[2:]
    t = t.reshape(N, C // 16, 16, *sp)
    perm = [0, 1] + list(range(3, 3 + len(sp))) + [2]
    return t.permute(*perm).contiguous().to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)


def unblocked_q(flat_u8, N, C, sp):
    t = flat_u8.view(torch.float8_e4m3fn).float().reshape(N, C // 16, *sp, 16)
    perm = [0, 1, 2 + len(sp)] + list(range(2, 2 + len(sp)))
    return t.permute(*perm).reshape(N, C, *sp)


@pytest.mark.parametrize('xf,yf', [(1, 1), (1, 0), (0, 1)])
@pytest.mark.parametrize('shape,cin,cout', [((4, 8, 16), 32, 32), ((9, 7, 17), 128, 64), ((16, 16, 16), 256, 256), ((8, 8, 8), 512, 128)])
def test_conv3_f8_e4m3_planes_exact_integers(nv, shape, cin, cout, xf, yf):
    """The K = 128 conv reading and / or writing e4m3 activation planes (include/iunet.h format 1): small integers, every product and
    sum exact -> bit equality with torch's conv; the e4m3 output is the 16-bit result rounded once more (values chosen inside the
    exactly representable range of both roundings: |y| <= 448 saturates, so the reference is clamped the same way)."""
    if not nv.lib().iunet_f8_pack_order(27, cin):
        pytest.skip('K16 operator order selected (IUNET_F8_K128=0)')
    g = torch.Generator().manual_seed(3)
    N, dtype, dev = 2, torch.bfloat16, 'cuda'
    D, H, W = shape
    vox = D * H * W
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cout, cin) + (3,) * 3, generator=g).float()
    w[torch.rand(w.shape, generator=g) < 0.9] = 0                                  # sparse: sums stay small
    w[:, 0, 1, 1, 1] = 1                                                           # (no all-zero output channel: the scale is defined)
    bias = torch.randint(-2, 3, (cout,), generator=g).float()
    want = torch.relu(F.conv3d(x, w, bias=bias, padding=1)).to(dtype).float()
    wd = w.contiguous().to(dev)
    dst = torch.zeros(nv.lib().iunet_f8_pack_conv3_bytes(cout, cin, 27), dtype=torch.uint8, device=dev)
    sc = torch.empty(cout, device=dev)
    nv.call('iunet_f8_pack_conv3', nv.ptr(wd), None, None, None, None, 1e-5, nv.ptr(dst), nv.ptr(sc), None, cout, cin, 27, nv.stream())
    xb = (blocked_q(x) if xf else blocked(x, dtype)).to(dev)
    y = torch.full((N * cout * vox,), 0x7f, dtype=torch.uint8, device=dev) if yf else torch.full((N * cout * vox,), float('nan'), dtype=dtype, device=dev)
    need = nv.lib().iunet_conv3_f8_workspace_elems(3, N, D, H, W, cin, cout)
    ws = torch.full((need,), float('nan'), device=dev) if need else None
    nv.call('iunet_conv3_f8_fwd_q', nv.DTYPE_CODE[dtype], 3, nv.ptr(xb), cin * vox, xf, nv.ptr(y), cout * vox, yf, nv.ptr(dst), nv.ptr(sc),
            nv.ptr(bias.to(dev)), N, D, H, W, cin, cout, 2, nv.ptr(ws), nv.stream())
    torch.cuda.synchronize()
    if yf:
        got = unblocked_q(y.cpu(), N, cout, shape)
        want = want.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    else:
        got = unblocked(y.float().cpu(), N, cout, shape)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), f'{(got != want).sum().item()} of {got.numel()} differ'


def test_e4m3_plane_producers_round_like_the_conv_loader(nv):
    """first conv, max-pool and transposed conv writing e4m3 planes: the bytes are the 16-bit kernels' results rounded to e4m3
    (round to nearest even, saturating), which is what the fp8 conv's loader makes of the 16-bit tensor."""
    g = torch.Generator().manual_seed(11)
    dev, dtype, dt = 'cuda', torch.bfloat16, nv.DTYPE_CODE[torch.bfloat16]
    N, D, H, W = 2, 4, 8, 16
    vox = D * H * W
    # ---- first conv (1 -> 64)
    x = torch.randint(0, 256, (N, 1, D, H, W), generator=g, dtype=torch.uint8).to(dev)
    w = (torch.randn((64, 1, 3, 3, 3), generator=g) * 3).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    wp = torch.empty(nv.lib().iunet_pack_first_conv_elems(64, 1, 27), dtype=dtype, device=dev)
    nv.call('iunet_pack_first_conv', dt, nv.ptr(w), None, nv.ptr(wp), 64, 1, 27, nv.stream())
    y16 = torch.empty(N * 64 * vox, dtype=dtype, device=dev)
    y8 = torch.empty(N * 64 * vox, dtype=torch.uint8, device=dev)
    strides = nv.ll_array((vox, vox, H * W, W, 1))
    nv.call('iunet_first_conv_fwd', dt, 3, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], strides, nv.ptr(y16), 64 * vox, nv.ptr(wp), nv.ptr(b), None,
            N, D, H, W, 1, 64, 1, nv.stream())
    nv.call('iunet_first_conv_fwd_q', dt, 3, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], strides, nv.ptr(y8), 64 * vox, nv.ptr(wp), nv.ptr(b),
            N, D, H, W, 1, 64, 1, nv.stream())
    torch.cuda.synchronize()
    ref = unblocked(y16.float().cpu(), N, 64, (D, H, W)).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert torch.equal(unblocked_q(y8.cpu(), N, 64, (D, H, W)), ref)
    # ---- max-pool of e4m3 planes = rounded max-pool
    t = (torch.randn((N, 32, D, H, W), generator=g) * 20).to(dtype).float()
    tq = blocked_q(t.clamp(-448, 448).to(torch.float8_e4m3fn).float()).to(dev)
    pq = torch.empty(N * 32 * vox // 8, dtype=torch.uint8, device=dev)
    nv.call('iunet_maxpool_q_fwd', 3, nv.ptr(tq), 32 * vox, nv.ptr(pq), 32 * vox // 8, 32, N, D // 2, H // 2, W // 2, nv.stream())
    torch.cuda.synchronize()
    ref = F.max_pool3d(t, 2).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert torch.equal(unblocked_q(pq.cpu(), N, 32, (D // 2, H // 2, W // 2)), ref)
    # ---- transposed conv (three kernels: direct, resident weights, chunked weights), output into the second half of a concat buffer
    for cin, cout, sp in ((64, 32, (2, 4, 16)), (64, 32, (4, 8, 16)), (256, 64, (4, 8, 16)), (32, 32, (1, 2, 3)),
                          (128, 64, (8, 16, 32)), (96, 32, (4, 16, 40)), (256, 64, (16, 16, 64))):      # resident weights: whole-granule epilogue (4 and 3 k-steps, ragged x; 8 k-steps on 8 waves)
        Di, Hi, Wi = sp
        vi, vo = Di * Hi * Wi, 8 * Di * Hi * Wi
        xi = (torch.randn((N, cin) + sp, generator=g)).to(dtype).float()
        wt = (torch.randn((cin, cout, 2, 2, 2), generator=g) * 0.5).to(dev)
        bt = torch.randn(cout, generator=g).to(dev)
        wpk = torch.empty(wt.numel(), dtype=dtype, device=dev)
        nv.call('iunet_pack_convT', dt, nv.ptr(wt), nv.ptr(wpk), cin, cout, 8, nv.stream())
        xb = blocked(xi, dtype).to(dev)
        o16 = torch.zeros(N * 2 * cout * vo, dtype=dtype, device=dev)
        o8 = torch.zeros(N * 2 * cout * vo, dtype=torch.uint8, device=dev)
        import ctypes
        nv.call('iunet_convT_fwd', dt, 3, nv.ptr(xb), cin * vi, ctypes.c_void_p(o16.data_ptr() + cout * vo * 2), 2 * cout * vo, nv.ptr(wpk),
                nv.ptr(bt), N, Di, Hi, Wi, cin, cout, nv.stream())
        nv.call('iunet_convT_fwd_q', dt, 3, nv.ptr(xb), cin * vi, ctypes.c_void_p(o8.data_ptr() + cout * vo), 2 * cout * vo, nv.ptr(wpk),
                nv.ptr(bt), N, Di, Hi, Wi, cin, cout, nv.stream())
        torch.cuda.synchronize()
        osp = (2 * Di, 2 * Hi, 2 * Wi)
        ref = unblocked(o16.float().cpu(), N, 2 * cout, osp).clamp(-448, 448).to(torch.float8_e4m3fn).float()
        got = unblocked_q(o8.cpu(), N, 2 * cout, osp)
        assert torch.equal(got[:, cout:], ref[:, cout:]), (cin, cout, sp)
        assert not got[:, :cout].any()                                             # the skip half of the buffer is untouched


def test_fp8_network_bits_do_not_depend_on_the_activation_format():
    """Engine with e4m3 planes between the fp8 convs (default) against the same engine with 16-bit tensors everywhere (IUNET_F8_Q=0):
    every producer rounds exactly as the consumer's loader would have -> identical logits, bit for bit."""
    import os, warnings
    from interactive_unet.unet import UNet
    p = unet_ref.init_params(dim=3, levels=3, base=32, ncls=3, seed=9, randomize_bn=True)
    x = torch.randint(0, 256, (2, 1, 16, 32, 32), dtype=torch.uint8, generator=torch.Generator().manual_seed(2)).cuda()
    outs = []
    for q in ('1', '0'):
        os.environ['IUNET_F8_Q'] = q
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                m = UNet(num_classes=3, dim=3, levels=3, base=32, act_dtype='bf16', pretrained=False, weight_dtype='fp8_e4m3')
            m.load_named(p)
            m = m.cuda().eval()
            eng = m.engine('eval')
            probs = m(x)
            assert eng.q_planes() == (q == '1')
            outs.append(probs.cpu())
        finally:
            os.environ.pop('IUNET_F8_Q', None)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('nd,shape,cin,cout', [
    (3, (4, 8, 16), 32, 32), (3, (6, 12, 20), 64, 64), (3, (9, 7, 17), 128, 32), (3, (8, 8, 16), 256, 64),   # 256: streamed weights
    (3, (16, 16, 16), 256, 256), (3, (2, 8, 16), 256, 32),                                                    # few tiles: half-size tile
    (3, (8, 8, 8), 512, 128), (3, (4, 8, 16), 1024, 64), (3, (8, 8, 16), 256, 32),                            # split along Cin
    (2, (16, 32), 32, 32), (2, (20, 70), 64, 32), (2, (48, 40), 256, 64), (2, (16, 32), 512, 32)])
def test_conv3_f8_exact_integers(nv, nd, shape, cin, cout, dtype):
    g = torch.Generator().manual_seed(1)
    N = 2
    x = torch.randint(-4, 5, (N, cin) + shape, generator=g).float()            # exact in e4m3 and in 16 bits
    w = torch.randint(-3, 4, (cout, cin) + (3,) * nd, generator=g).float()      # x 2^k: exact in e4m3
    w[:, :, (0,) * nd] += 0                                                      # (asymmetric by construction: random)
    bias = torch.randint(-8, 9, (cout,), generator=g).float()
    conv = F.conv3d if nd == 3 else F.conv2d
    want = torch.relu(conv(x, w, bias=bias, padding=1))
    got, _, sc, _ = run_conv_f8(nv, x, w, DT[dtype], nd, epi=2, bias=bias)
    assert torch.isfinite(got).all()
    # every product and every fp32 sum is exact (|sums| < 2^24); the one rounding is the store to 16 bits, round to nearest even
    assert torch.equal(got, want.to(DT[dtype]).float()), f'{(got != want.to(DT[dtype]).float()).sum().item()} of {got.numel()} differ'
    D, H, W = shape if nd == 3 else (1,) + shape
    if nv.lib().iunet_conv3_f8_workspace_elems(nd, N, D, H, W, cin, cout):      # a split-K shape: the unsplit launch gives the same bits
        got1, _, _, _ = run_conv_f8(nv, x, w, DT[dtype], nd, epi=2, bias=bias, splitk=False)
        assert torch.equal(got1, got)


@pytest.mark.parametrize('nd', [2, 3])
def test_conv3_f8_rounds_activations_like_the_oracle(nv, nd):
    """Inputs that are not e4m3 values (and some beyond 448): the result equals the fp32 conv of the ORACLE's rounding of
    the same input (quantize_act_e4m3: nearest even, saturating) -- sums of exact products, compared with a tolerance of
    one fp32 accumulation order."""
    g = torch.Generator().manual_seed(5)
    cin, cout = 64, 32
    shape = (4, 8, 16) if nd == 3 else (16, 32)
    x = (torch.randn((1, cin) + shape, generator=g) * 3).to(torch.bfloat16).float()
    x.view(-1)[:7] = torch.tensor([500.0, -1000.0, 448.0, 0.0009765625, -0.0029296875, 17.0, 0.0])   # saturation, subnormals, ties
    w = torch.randint(-3, 4, (cout, cin) + (3,) * nd, generator=g).float()
    conv = F.conv3d if nd == 3 else F.conv2d
    xq = unet_ref.quantize_act_e4m3(x)
    assert xq.abs().max() == 448 and not torch.equal(xq, x)
    want = conv(xq, w, padding=1)
    got, _, _, _ = run_conv_f8(nv, x, w, torch.bfloat16, nd, epi=0)
    want_b = want.to(torch.bfloat16).float()
    # the device rounds its fp32 sum to bf16 once; the sums themselves differ by the order of fp32 additions at most
    assert (got - want_b).abs().max() <= 2 ** -7 * want.abs().max()
    assert ((got - want_b).abs() > 0).float().mean() < 0.02


def _f8_net_case(dim, levels, base, ncls, shape, seed):
    import warnings
    from interactive_unet.unet import UNet
    from interactive_unet.engine import F8Conv
    from scipy import ndimage
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, ncls=ncls, seed=seed, randomize_bn=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=ncls, dim=dim, levels=levels, base=base, act_dtype='bf16', pretrained=False,
                 weight_dtype='fp8_e4m3')
    m.load_named(p)
    m = m.cuda().eval()
    eng = m.engine('eval')
    assert isinstance(eng.packed['enc0.conv2'][0], F8Conv) and isinstance(eng.packed['dec0.conv1'][0], F8Conv)
    assert eng.packed['dec0.conv1'][0].bytes.dtype == torch.uint8              # one byte per weight in HBM
    rng = np.random.default_rng(5)
    v = ndimage.gaussian_filter(rng.random(shape), 3)
    x = torch.tensor(((v - v.min()) / (v.max() - v.min()) * 255).astype(np.uint8))[None, None]
    probs = m(x.cuda()).cpu()
    xf = x.float() / 255.0
    kw = dict(dim=dim, levels=levels, act_dtype=torch.bfloat16, weight_quant='fp8_e4m3')
    ref_q = unet_ref.forward(p, xf, act_quant=True, **kw)                       # same quantised operands as the device
    ref_w = unet_ref.forward(p, xf, **kw)                                        # weights only (the round-1 fake-quant path)
    ref_f = unet_ref.forward(p, xf, dim=dim, levels=levels)                      # fp32
    d = (probs - ref_q).abs()
    r = dict(mean=d.mean().item(), max=d.max().item(), agree=(probs.argmax(1) == ref_q.argmax(1)).float().mean().item(),
             dev_vs_f32=(probs - ref_f).abs().mean().item(), emu_vs_f32=(ref_q - ref_f).abs().mean().item(),
             dev_vs_wonly=(probs - ref_w).abs().mean().item())
    print(f'{dim}-D {levels} levels base {base} on fp8 MFMA: vs same-operand emulation mean |dp| = {r["mean"]:.2e}, max = {r["max"]:.2e}, '
          f'class map equal on {100 * r["agree"]:.2f} %; mean |dp| vs fp32: device {r["dev_vs_f32"]:.2e}, emulation {r["emu_vs_f32"]:.2e}')
    return r


def test_shallow_network_on_fp8_matrix_cores_tracks_the_emulation():
    """Two levels (6 stage convs): little depth for a differently rounded value to spread -- the device and the CPU emulation
    with the same quantised operands agree closely."""
    for dim, shape in ((3, (16, 32, 48)), (2, (64, 96))):
        r = _f8_net_case(dim, 2, 32, 3, shape, seed=7)
        assert r['mean'] <= 2e-3 and r['agree'] >= 0.995, r
        assert r['mean'] < r['dev_vs_wonly']                  # the activation rounding is really applied


def test_c5_network_on_fp8_matrix_cores():
    """BASELINE.json configs[4]: 3-D, 5 levels, base 64, 4 classes; stage convs on the fp8 MFMA.  Checker = the CPU emulation
    with the SAME quantised operands (e4m3 weights with the same scales, activations rounded to bf16 in memory and to e4m3
    in front of every stage conv).  Tolerance (stated, not 1e-3: SURVEY 8d "fp8 is not held to 1e-3 vs fp32"): an e4m3
    rounding step is 2^-4 relative; a value near a rounding boundary rounds differently after a different fp32 summation
    order, and 22 quantising layers spread that -- the device and the emulation are two equally valid evaluations of the
    same quantised network, as far from each other as each is from fp32 (measured: mean |dp| 1.1e-2 between them, 1.5e-2
    emulation vs fp32; this random-init 4-class net has near-uniform probabilities, so 4 % of the argmax flips).  Asserted:
    mean |dp| <= 2e-2, max <= 0.3, class map equal on >= 93 %, and the device is no further from fp32 than 1.3 x the
    emulation is."""
    r = _f8_net_case(3, 5, 64, 4, (32, 32, 48), seed=6)
    assert r['mean'] <= 2e-2 and r['max'] <= 0.3 and r['agree'] >= 0.93, r
    assert r['dev_vs_f32'] <= 1.3 * r['emu_vs_f32'], r


def test_weights_only_mode_w8a16():
    """UNet(weight_dtype='fp8_e4m3', act_quant=False): BASELINE C5's literal "fp8 weights / bf16 activations" -- e4m3-valued
    operators (per-channel power-of-two scales) on the 16-bit matrix cores, activations NOT quantised.  Checker: the oracle
    with quantised weights only."""
    import warnings
    from interactive_unet.unet import UNet
    from interactive_unet.engine import F8Conv
    from scipy import ndimage
    dim, levels, base, ncls, shape = 3, 2, 32, 3, (16, 32, 48)
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, ncls=ncls, seed=7, randomize_bn=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=ncls, dim=dim, levels=levels, base=base, act_dtype='bf16', pretrained=False,
                 weight_dtype='fp8_e4m3', act_quant=False)
    m.load_named(p)
    m = m.cuda().eval()
    assert not isinstance(m.engine('eval').packed['dec0.conv1'][0], F8Conv)
    rng = np.random.default_rng(5)
    v = ndimage.gaussian_filter(rng.random(shape), 3)
    x = torch.tensor(((v - v.min()) / (v.max() - v.min()) * 255).astype(np.uint8))[None, None]
    probs = m(x.cuda()).cpu()
    xf = x.float() / 255.0
    kw = dict(dim=dim, levels=levels, act_dtype=torch.bfloat16, weight_quant='fp8_e4m3')
    ref_w = unet_ref.forward(p, xf, **kw)
    ref_q = unet_ref.forward(p, xf, act_quant=True, **kw)
    d = (probs - ref_w).abs()
    print(f'W8A16: mean |dp| vs weights-only oracle {d.mean().item():.2e}, max {d.max().item():.2e}; vs W8A8 emulation {(probs - ref_q).abs().mean().item():.2e}')
    assert d.mean().item() <= 1e-3 and d.max().item() <= 3e-2
    assert d.mean().item() < (probs - ref_q).abs().mean().item()


def test_c5_at_128_cubed_against_the_fp32_mode():
    """BASELINE.json configs[4] at its full tile (one 128^3 chunk, 5 levels, base 64, 4 classes): the fp8 matrix-core forward
    (W8A8) and the weights-only mode (W8A16) against the native fp32 mode with the UNQUANTISED operators -- the device-side
    stand-in for the CPU oracle at a size it cannot finish in a test.  fp8 is not held to 1e-3: the figures are
    printed; asserted are the bounds measured at 32 x 32 x 48 against the CPU (mean |dp| <= 2.5e-2, class map >= 90 %), and
    that quantising the activations too costs accuracy in the expected direction."""
    import warnings
    from interactive_unet.unet import UNet
    from interactive_unet.engine_f32 import EngineF32
    dim, levels, base, ncls, S = 3, 5, 64, 4, 128
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, ncls=ncls, seed=6, randomize_bn=True)
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand((1, 1, S // 4, S // 4, S // 4), generator=g, device='cuda')
    x = torch.nn.functional.interpolate(x, size=(S, S, S), mode='trilinear', align_corners=False)
    x = ((x - x.amin()) / (x.amax() - x.amin()) * 254 + 1).to(torch.uint8)
    vox = S ** 3
    e32 = EngineF32(dim, levels, base, 1, ncls, 'cuda')
    e32.load_eval({k: v.cuda() for k, v in p.items()})
    ref = torch.empty((1, ncls, S, S, S), device='cuda')
    e32.infer(x, (vox, vox, S * S, S, 1), 1, S, S, S, probs=ref)
    del e32
    res = {}
    for name, aq in (('W8A8', True), ('W8A16', False)):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(num_classes=ncls, dim=dim, levels=levels, base=base, act_dtype='bf16', pretrained=False,
                     weight_dtype='fp8_e4m3', act_quant=aq)
        m.load_named(p)
        probs = m.cuda().eval()(x)
        d = (probs - ref).abs()
        res[name] = (d.mean().item(), d.max().item(), (probs.argmax(1) == ref.argmax(1)).float().mean().item())
        print(f'C5 @ 128^3 {name} vs fp32 mode: mean |dp| = {res[name][0]:.2e}, max = {res[name][1]:.2e}, class map equal on {100 * res[name][2]:.2f} %')
        del m
        torch.cuda.empty_cache()
    assert res['W8A8'][0] <= 2.5e-2 and res['W8A8'][2] >= 0.90, res
    assert res['W8A16'][0] <= res['W8A8'][0], res
