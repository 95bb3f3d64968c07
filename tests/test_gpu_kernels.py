"""Per-kernel parity of the HIP path (through the C ABI) against plain torch fp32 ops on
the CPU and the oracle.  Needs an MI355X: run with -m gpu."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'f16': torch.float16, 'bf16': torch.bfloat16}


@pytest.fixture(scope='module')
def nv():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from interactive_unet import _native
    _native.lib()
    return _native


def blocked(t, dtype):
    N, C = t.shape[:2]
    sp = t.shape[2:]
    t = t.reshape(N, C // 8, 8, *sp)
    perm = [0, 1] + list(range(3, 3 + len(sp))) + [2]
    return t.permute(*perm).contiguous().to(dtype).reshape(-1)


def unblocked(flat, N, C, sp):
    t = flat.reshape(N, C // 8, *sp, 8)
    perm = [0, 1, 2 + len(sp)] + list(range(2, 2 + len(sp)))
    return t.permute(*perm).reshape(N, C, *sp)


def run_conv3(nv, x, w, dtype, nd, scale=None, bias=None, epi=0, stats=False, mode=0, layout=None):
    """x [N,Cin,D,H,W] fp32 cpu, w [Cout,Cin,k..] fp32 cpu -> y [N,Cout',D,H,W] fp32 cpu."""
    dev = 'cuda'
    N, Cin = x.shape[:2]
    sp = tuple(x.shape[2:])
    D, H, W = sp if nd == 3 else (1,) + sp
    taps = 3 ** nd
    Co_p, Ci_p = (w.shape[0], w.shape[1]) if mode == 0 else (w.shape[1], w.shape[0])
    xb = blocked(x, dtype).to(dev)
    wd = w.contiguous().to(dev)
    if layout is None:
        layout = nv.lib().iunet_conv3_pick_layout(nd, N, D, H, W, Ci_p, Co_p)
    pmode = mode | (6 if layout == 3 else 2 if layout in (1, 2) else 0)      # layout 3: the compact K16 order (mode bit 2)
    wpk = torch.empty(nv.pack_conv3_elems(w.shape[0], w.shape[1], taps, pmode), dtype=dtype, device=dev)
    sc = None if scale is None else scale.to(dev)
    nv.call('iunet_pack_conv3', nv.DTYPE_CODE[dtype], nv.ptr(wd), nv.ptr(sc), nv.ptr(wpk), w.shape[0], w.shape[1],
            taps, pmode, nv.stream())
    vox = D * H * W
    y = torch.full((N * Co_p * vox,), float('nan'), dtype=dtype, device=dev)
    bd = None if bias is None else bias.to(dev)
    st = None
    if stats:
        nt = nv.lib().iunet_conv3_stats_parts(nd, N, D, H, W, Co_p, layout)
        st = torch.full((nt * Co_p * 2,), float('nan'), dtype=torch.float32, device=dev)     # every row must be written
    nv.call('iunet_conv3_fwd', nv.DTYPE_CODE[dtype], nd, nv.ptr(xb), Ci_p * vox, nv.ptr(y), Co_p * vox, nv.ptr(wpk),
            nv.ptr(bd), nv.ptr(st), N, D, H, W, Ci_p, Co_p, epi, layout, nv.stream())
    torch.cuda.synchronize()
    out = unblocked(y.float().cpu(), N, Co_p, sp)
    if stats:
        return out, st.cpu().reshape(-1, Co_p, 2).sum(0)
    return out


@pytest.mark.parametrize('nd,shape,cin,cout', [
    (2, (16, 32), 32, 32), (2, (48, 40), 64, 64), (2, (20, 70), 64, 32), (2, (16, 16), 32, 128),
    (3, (4, 8, 16), 32, 32), (3, (6, 12, 20), 64, 64), (3, (8, 8, 8), 32, 64), (3, (20, 24, 40), 32, 32),
    (3, (12, 16, 48), 64, 32), (3, (9, 7, 17), 32, 32)])
def test_conv3_exact_integers(nv, nd, shape, cin, cout):
    """Small-integer data: every product and sum is exact in f16 x f16 -> f32, so the MFMA
    path must equal the fp32 reference bit for bit (catches any fragment-map / tap / halo
    error; weights are asymmetric in every axis)."""
    g = torch.Generator().manual_seed(1)
    N = 2
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cout, cin) + (3,) * nd, generator=g).float()
    ref = (F.conv2d if nd == 2 else F.conv3d)(x, w, padding=1)
    for dt in (torch.float16, torch.bfloat16):
        layouts = ((0, 1) if cout % 64 == 0 and cin % 32 == 0 else (1,)) + (2,)
        for layout in layouts:                                            # every kernel structure that is legal
            got = run_conv3(nv, x, w, dt, nd, layout=layout)
            ok = ref.abs() <= (2048 if dt == torch.float16 else 256)   # exactly representable outputs
            assert torch.equal(got[ok], ref[ok]), (dt, layout, (got - ref)[ok].abs().max())


@pytest.mark.parametrize('shape,cin,cout', [((58, 62, 120), 64, 32), ((32, 32, 64), 128, 64)])
def test_conv3_tile_pairs_exact_integers(nv, shape, cin, cout):
    """Grids on which the layout-2 kernel walks its tiles in pairs (one weight stream per two tiles, two accumulator sets;
    iunet_conv3_tile_pairs): ragged on every axis, forward with statistics and the data gradient, bit for bit."""
    g = torch.Generator().manual_seed(21)
    N = 2
    assert nv.lib().iunet_conv3_tile_pairs(3, N, *shape, cin, cout) == 1
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3, 3), generator=g).float()
    ref = F.conv3d(x, w, padding=1)
    for dt in (torch.float16, torch.bfloat16):
        got, st = run_conv3(nv, x, w, dt, 3, layout=2, stats=True)
        ok = ref.abs() <= (2048 if dt == torch.float16 else 256)
        assert torch.equal(got[ok], ref[ok]), (dt, (got - ref)[ok].abs().max())
        assert torch.equal(st[:, 0], ref.sum((0, 2, 3, 4)))                 # integer sums below 2^24: exact in fp32 in any order
    if nv.lib().iunet_conv3_tile_pairs(3, N, *shape, cout, cin) == 1:       # the data gradient: roles of Cin / Cout swapped
        dy = torch.randint(-2, 3, (N, cout) + shape, generator=g).float()
        want = F.conv_transpose3d(dy, w, padding=1)
        got = run_conv3(nv, dy, w, torch.float16, 3, layout=2, mode=1)
        ok = want.abs() <= 2048
        assert torch.equal(got[ok], want[ok])


@pytest.mark.parametrize('shape,cin,cout', [((58, 62, 120), 64, 32), ((32, 32, 64), 128, 64), ((20, 24, 40), 64, 64),
                                            ((4, 8, 16), 64, 32), ((6, 9, 17), 256, 64)])         # the last two: the half-size tile
def test_conv3_compact_operator_exact_integers(nv, shape, cin, cout):
    """Layout 3: the compact K16 order and the padding-free step (the ninth filter column of two consecutive 16-channel chunks in
    one k-slot, three halo buffers): forward with statistics and the data gradient, bit for bit."""
    g = torch.Generator().manual_seed(22)
    N = 2
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3, 3), generator=g).float()
    ref = F.conv3d(x, w, padding=1)
    for dt in (torch.float16, torch.bfloat16):
        got, st = run_conv3(nv, x, w, dt, 3, layout=3, stats=True)
        ok = ref.abs() <= (2048 if dt == torch.float16 else 256)
        assert torch.equal(got[ok], ref[ok]), (dt, (got - ref)[ok].abs().max())
        assert torch.equal(st[:, 0], ref.sum((0, 2, 3, 4)))
    if cout > 32:
        dy = torch.randint(-2, 3, (N, cout) + shape, generator=g).float()
        want = F.conv_transpose3d(dy, w, padding=1)
        got = run_conv3(nv, dy, w, torch.float16, 3, layout=3, mode=1)
        ok = want.abs() <= 2048
        assert torch.equal(got[ok], want[ok])


@pytest.mark.parametrize('shape,cin,cout', [((16, 32), 32, 32), ((48, 40), 64, 64), ((20, 70), 128, 64), ((33, 65), 256, 32),
                                            ((70, 130), 64, 128)])
def test_conv2_cross_pair_exact_integers(nv, shape, cin, cout):
    """Layout 3 in 2-D: the compact order of the 3^2 filter and the cross-pair step (the third filter column of a step's two 16-channel
    halves in one k-group; resident weights up to 64 input channels, streamed by LDS-DMA beyond): forward with statistics, bias +
    ReLU epilogue and the data gradient, bit for bit; the batched packer (descriptor kind 6) writes the per-layer kernel's bytes."""
    if os.environ.get('IUNET_NO_COMPACT2D'):
        pytest.skip('A/B switch IUNET_NO_COMPACT2D: no layout 3 in 2-D')
    g = torch.Generator().manual_seed(23)
    N = 3
    assert nv.lib().iunet_conv3_compact_ok(2, N, 1, *shape, cin, cout, 0, 0) == 1
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float()
    bias = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = F.conv2d(x, w, padding=1)
    for dt in (torch.float16, torch.bfloat16):
        got, st = run_conv3(nv, x, w, dt, 2, layout=3, stats=True)
        ok = ref.abs() <= (2048 if dt == torch.float16 else 256)
        assert torch.equal(got[ok], ref[ok]), (dt, (got - ref)[ok].abs().max())
        assert torch.equal(st[:, 0], ref.sum((0, 2, 3)))
    refb = F.relu(ref + bias.view(1, -1, 1, 1))
    got = run_conv3(nv, x, w, torch.float16, 2, layout=3, bias=bias, epi=2)
    ok = refb.abs() <= 2048
    assert torch.equal(got[ok], refb[ok])
    dy = torch.randint(-2, 3, (N, cout) + shape, generator=g).float()
    want = F.conv_transpose2d(dy, w, padding=1)
    got = run_conv3(nv, dy, w, torch.float16, 2, layout=3, mode=1)
    ok = want.abs() <= 2048
    assert torch.equal(got[ok], want[ok])
    # descriptor kind 6 of iunet_pack_batch == iunet_pack_conv3 mode 6 (forward and data-gradient operators)
    wd = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    for dg in (0, 1):
        n = nv.pack_conv3_elems(cout, cin, 9, 6 | dg)
        a = torch.zeros(n, dtype=torch.float16, device='cuda')
        b = torch.ones(n, dtype=torch.float16, device='cuda')
        nv.call('iunet_pack_conv3', 0, nv.ptr(wd), None, nv.ptr(a), cout, cin, 9, 6 | dg, nv.stream())
        nv.PackTable([nv.make_desc(wd, b, cout, cin, 9, 6, torch.float16, dg)], 'cuda', sources=[wd]).run()
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), dg


@pytest.mark.parametrize('nd', [2, 3])
def test_conv3_random_bias_relu_stats(nv, nd):
    g = torch.Generator().manual_seed(2)
    shape = (40, 48) if nd == 2 else (8, 16, 24)
    cin, cout, N = 64, 64, 2
    x = torch.randn((N, cin) + shape, generator=g).half().float()
    w = torch.randn((cout, cin) + (3,) * nd, generator=g) * (2.0 / (cin * 3 ** nd)) ** 0.5
    scale = 0.5 + torch.rand(cout, generator=g)
    bias = torch.randn(cout, generator=g) * 0.1
    wf = (w * scale.view(-1, *([1] * (nd + 1)))).half().float()
    conv = F.conv2d if nd == 2 else F.conv3d
    ref = F.relu(conv(x, wf, bias=bias, padding=1))
    got = run_conv3(nv, x, w, torch.float16, nd, scale=scale, bias=bias, epi=2)
    assert (got - ref).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())
    rr = conv(x, w.half().float(), padding=1)
    dims = [0] + list(range(2, 2 + nd))
    for layout in (None, 1, 2):
        raw, st = run_conv3(nv, x, w, torch.float16, nd, stats=True, layout=layout)
        assert torch.allclose(st[:, 0], rr.sum(dims), rtol=1e-3, atol=1e-1), layout
        assert torch.allclose(st[:, 1], (rr * rr).sum(dims), rtol=1e-3, atol=1e-1), layout
        assert (raw - rr).abs().max() <= 2e-3 * rr.abs().max().item(), layout
    if nd == 3:      # the weight-stationary structure with the 8 x 8 x 16 tile (Cin 32): same statistics grid
        x2, w2 = x[:, :32].contiguous(), w[:32, :32].contiguous()
        r2 = conv(x2, w2.half().float(), padding=1)
        for layout in (1, 2):
            raw, st = run_conv3(nv, x2, w2, torch.float16, nd, stats=True, layout=layout)
            assert torch.allclose(st[:, 0], r2.sum(dims), rtol=1e-3, atol=1e-1), layout
            assert torch.allclose(st[:, 1], (r2 * r2).sum(dims), rtol=1e-3, atol=1e-1), layout


@pytest.mark.parametrize('nd', [2, 3])
def test_conv3_dgrad_mode(nv, nd):
    """mode-1 packing turns the same kernel into the data gradient."""
    g = torch.Generator().manual_seed(3)
    shape = (24, 40) if nd == 2 else (4, 8, 24)
    cin, cout = 64, 32
    w = torch.randint(-1, 2, (cout, cin) + (3,) * nd, generator=g).float()
    dy = torch.randint(-2, 3, (1, cout) + shape, generator=g).float()
    convT = F.conv_transpose2d if nd == 2 else F.conv_transpose3d
    ref = convT(dy, w, padding=1)                     # = d/dx of conv(x, w, padding=1) . dy
    got = run_conv3(nv, dy, w, torch.float16, nd, mode=1)
    ok = ref.abs() <= 2048
    assert torch.equal(got[ok], ref[ok])


@pytest.mark.parametrize('nd,in_dtype', [(2, torch.uint8), (2, torch.float32), (3, torch.uint8), (3, torch.float16)])
def test_first_conv(nv, nd, in_dtype):
    g = torch.Generator().manual_seed(4)
    shape = (24, 36) if nd == 2 else (6, 10, 12)
    N, cin, cout = 2, 1, 32
    taps = 3 ** nd
    if in_dtype == torch.uint8:
        xi = torch.randint(0, 256, (N, cin) + shape, generator=g, dtype=torch.uint8)
        xf = (xi.float() / 255.0).half().float()
    else:
        xi = torch.rand((N, cin) + shape, generator=g).to(in_dtype)
        xf = xi.float().half().float()
    w = torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.3
    scale = 0.5 + torch.rand(cout, generator=g)
    bias = torch.randn(cout, generator=g) * 0.1
    wf = (w * scale.view(-1, *([1] * (nd + 1)))).half().float()
    ref = F.relu((F.conv2d if nd == 2 else F.conv3d)(xf, wf, bias=bias, padding=1))
    dev = 'cuda'
    xd = xi.to(dev)
    wd, sd, bd = w.to(dev), scale.to(dev), bias.to(dev)
    wp = torch.empty(nv.lib().iunet_pack_first_conv_elems(cout, cin, taps), dtype=torch.float16, device=dev)
    nv.call('iunet_pack_first_conv', 0, nv.ptr(wd), nv.ptr(sd), nv.ptr(wp), cout, cin, taps, nv.stream())
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    y = torch.empty(N * cout * vox, dtype=torch.float16, device=dev)
    nb = nv.lib().iunet_conv3_num_tiles(nd, N, D, H, W)
    st = torch.zeros(nb * cout * 2, dtype=torch.float32, device=dev)
    strides = (cin * vox, vox, H * W, W, 1)
    nv.call('iunet_first_conv_fwd', 0, nd, nv.ptr(xd), nv.IN_DTYPE_CODE[in_dtype], nv.ll_array(strides), nv.ptr(y),
            cout * vox, nv.ptr(wp), nv.ptr(bd), nv.ptr(st), N, D, H, W, cin, cout, 1, nv.stream())
    torch.cuda.synchronize()
    got = unblocked(y.float().cpu(), N, cout, shape)
    assert (got - ref).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())
    raw = (F.conv2d if nd == 2 else F.conv3d)(xf, wf, padding=1)
    s = st.cpu().reshape(-1, cout, 2).sum(0)
    dims = [0] + list(range(2, 2 + nd))
    assert torch.allclose(s[:, 0], raw.sum(dims), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[:, 1], (raw * raw).sum(dims), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize('nd', [2, 3])
@pytest.mark.parametrize('dt', ['f16', 'bf16'])
def test_maxpool(nv, nd, dt):
    dtype = DT[dt]
    g = torch.Generator().manual_seed(5)
    shape = (12, 20) if nd == 2 else (4, 6, 10)
    N, C = 2, 32
    x = torch.randn((N, C) + shape, generator=g).to(dtype).float()
    ref = (F.max_pool2d if nd == 2 else F.max_pool3d)(x, 2)
    xb = blocked(x, dtype).cuda()
    osp = tuple(s // 2 for s in shape)
    Do, Ho, Wo = osp if nd == 3 else (1,) + osp
    y = torch.empty(N * C * Do * Ho * Wo, dtype=dtype, device='cuda')
    vin = int(np.prod(shape))
    nv.call('iunet_maxpool_fwd', nv.DTYPE_CODE[dtype], nd, nv.ptr(xb), C * vin, nv.ptr(y), C * Do * Ho * Wo, C, N,
            Do, Ho, Wo, nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(unblocked(y.float().cpu(), N, C, osp), ref)


@pytest.mark.parametrize('nd,shape,cin,cout', [(2, (8, 16), 64, 32), (2, (6, 20), 256, 128), (3, (4, 4, 16), 64, 32),
                                               (3, (8, 16, 24), 64, 32), (2, (40, 72), 128, 64), (3, (6, 10, 40), 96, 64),
                                               (3, (2, 3, 8), 128, 64),
                                               (3, (16, 32, 80), 64, 32), (3, (16, 16, 72), 128, 64), (2, (96, 200), 96, 32),      # resident weights, several iterations per wave, ragged x
                                               (3, (16, 16, 72), 256, 64), (2, (64, 72), 256, 64),      # Cin = 256 resident (3-D, >= 2048 voxel groups: 8 waves on one copy)
                                               (3, (8, 8, 16), 256, 128), (3, (5, 9, 20), 512, 64), (2, (24, 40), 256, 32)])     # Cin > 128: weights chunked through LDS
def test_convT_exact_integers(nv, nd, shape, cin, cout):
    g = torch.Generator().manual_seed(6)
    N = 2
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    w = torch.randint(-1, 2, (cin, cout) + (2,) * nd, generator=g).float()
    bias = torch.randint(-3, 4, (cout,), generator=g).float()
    ref = (F.conv_transpose2d if nd == 2 else F.conv_transpose3d)(x, w, bias=bias, stride=2)
    D, H, W = shape if nd == 3 else (1,) + shape
    osp = tuple(2 * s for s in shape)
    xb = blocked(x, torch.float16).cuda()
    wd, bd = w.cuda(), bias.cuda()
    wpk = torch.empty(w.numel(), dtype=torch.float16, device='cuda')
    nv.call('iunet_pack_convT', 0, nv.ptr(wd), nv.ptr(wpk), cin, cout, 2 ** nd, nv.stream())
    vin, vout = int(np.prod(shape)), int(np.prod(osp))
    y = torch.full((N * cout * vout,), float('nan'), dtype=torch.float16, device='cuda')
    nv.call('iunet_convT_fwd', 0, nd, nv.ptr(xb), cin * vin, nv.ptr(y), cout * vout, nv.ptr(wpk), nv.ptr(bd),
            N, D, H, W, cin, cout, nv.stream())
    torch.cuda.synchronize()
    got = unblocked(y.float().cpu(), N, cout, osp)
    ok = ref.abs() <= 2048
    assert torch.equal(got[ok], ref[ok])


@pytest.mark.parametrize('ncls', [2, 3, 10])
def test_head_softmax_argmax(nv, ncls):
    g = torch.Generator().manual_seed(7)
    N, C0, shape = 2, 32, (3, 10, 12)
    x = torch.randn((N, C0) + shape, generator=g).half().float()
    w = torch.randn(ncls, C0, generator=g) * 0.3
    b = torch.randn(ncls, generator=g) * 0.1
    ref_l = F.conv3d(x, w.view(ncls, C0, 1, 1, 1), bias=b)
    ref_p = torch.softmax(ref_l, 1)
    D, H, W = shape
    vox = D * H * W
    xb = blocked(x, torch.float16).cuda()
    logits = torch.empty(N, ncls, D, H, W, device='cuda')
    probs = torch.empty(N, ncls, D, H, W, device='cuda')
    cls = torch.empty(N, vox, dtype=torch.uint8, device='cuda')
    wd, bd = w.cuda(), b.cuda()          # keep the device copies alive across the async launch
    nv.call('iunet_head_fwd', 0, nv.ptr(xb), C0 * vox, C0, nv.ptr(wd), nv.ptr(bd), ncls, nv.ptr(logits),
            nv.ptr(probs), nv.ptr(cls), nv.ll_array((ncls * vox, vox, H * W, W, 1)), 1.0, 0, N, D, H, W, nv.stream())
    torch.cuda.synchronize()
    assert (logits.cpu() - ref_l).abs().max() < 1e-5
    assert (probs.cpu() - ref_p).abs().max() < 1e-6
    # class map must be exactly np.argmax of the probabilities we returned (predict.py:38)
    want = np.argmax(probs.cpu().numpy(), axis=1).reshape(N, vox)
    assert np.array_equal(cls.cpu().numpy(), want)


@pytest.mark.parametrize('dt', ['f16', 'bf16'])
def test_pack_batch_matches_per_layer_packs(nv, dt):
    """iunet_pack_batch (one launch, in-kernel BatchNorm fold) is bit-identical to the per-layer pack entry
    points fed with the (CPU, correctly rounded) fold scale = gamma / sqrt(var + eps), bias = beta - mean * scale."""
    T, code, dev = DT[dt], nv.DTYPE_CODE[DT[dt]], 'cuda'
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    s = nv.stream()
    descs, expect, keep = [], [], []

    def bn_of(c):
        bn = [torch.randn(c, generator=g), torch.randn(c, generator=g), torch.randn(c, generator=g),
              torch.rand(c, generator=g) + 0.1]
        n = [t.numpy() for t in bn]                            # numpy fp32: every operation correctly rounded
        scale = n[0] / np.sqrt(n[3] + np.float32(1e-5))         # (torch's CPU sqrt/divide differ from it by an ulp)
        bias = n[1] - n[2] * scale
        return [t.to(dev) for t in bn], torch.from_numpy(scale).to(dev), torch.from_numpy(bias).to(dev)

    for taps in (9, 27):
        for cout, cin in ((32, 64), (64, 32), (128, 64)):
            w = rnd(cout, cin, taps)
            bn, scale, bias = bn_of(cout)
            for dg in (0, 1):
                pc = nv.PackedConv(cout, cin, taps, T, dev, dgrad=bool(dg))
                pc.pack(w, None if dg else scale)
                ref = {lay: b.clone() for lay, b in pc.buf.items()}
                for b in pc.buf.values():
                    b.zero_()
                bias_out = torch.zeros(cout, device=dev)
                ds = pc.descs(w, None if dg else bn, None if dg else bias_out)
                descs += ds
                for lay in sorted(pc.buf, reverse=True):
                    expect.append((f'conv3 taps{taps} {cout}x{cin} dg{dg} layout{lay}', pc.buf[lay], ref[lay]))
                if not dg:
                    expect.append((f'bias taps{taps} {cout}x{cin}', bias_out, bias))
                keep += [w, bn, pc]
        # first conv (Cin 1..4) with fold
        for cin in (1, 3):
            w = rnd(32, cin, taps)
            bn, scale, bias = bn_of(32)
            n = nv.lib().iunet_pack_first_conv_elems(32, cin, taps)
            ref = torch.zeros(n, dtype=T, device=dev)
            nv.call('iunet_pack_first_conv', code, nv.ptr(w), nv.ptr(scale), nv.ptr(ref), 32, cin, taps, s)
            dst, bias_out = torch.zeros(n, dtype=T, device=dev), torch.zeros(32, device=dev)
            descs.append(nv.make_desc(w, dst, 32, cin, taps, 2, T, bn=bn, bias_out=bias_out))
            expect += [(f'first taps{taps} cin{cin}', dst, ref), (f'first bias taps{taps} cin{cin}', bias_out, bias)]
            keep += [w, bn]
    for npos in (4, 8):
        for cin, cout in ((64, 32), (256, 128)):
            w = rnd(cin, cout, npos)
            for kind, fn in ((3, 'iunet_pack_convT'), (4, 'iunet_pack_convT_dgrad')):
                ref = torch.zeros(w.numel(), dtype=T, device=dev)
                nv.call(fn, code, nv.ptr(w), nv.ptr(ref), cin, cout, npos, s)
                dst = torch.zeros(w.numel(), dtype=T, device=dev)
                descs.append(nv.make_desc(w, dst, cout, cin, npos, kind, T))
                expect.append((f'{fn} npos{npos} {cin}->{cout}', dst, ref))
            keep.append(w)
    table = nv.PackTable(descs, dev)
    table.run()
    torch.cuda.synchronize()
    for name, got, ref in expect:
        assert got.dtype == ref.dtype and got.shape == ref.shape, name
        bits = torch.int16 if got.dtype != torch.float32 else torch.int32
        assert torch.equal(got.view(bits), ref.view(bits)), name


def test_pack_batch_e4m3_quantisation_matches_oracle(nv):
    """Device-side e4m3 weight quantisation (in-kernel BatchNorm fold -> per-output-channel scale -> round) packs the
    same bits as packing the oracle's quantised weights unquantised."""
    from oracle import unet_ref
    T, dev = torch.bfloat16, 'cuda'
    code = nv.DTYPE_CODE[T]
    g = torch.Generator().manual_seed(12)
    descs, expect, keep = [], [], []
    for taps, cout, cin in ((27, 64, 32), (9, 32, 64), (27, 32, 32)):
        w = torch.randn(cout, cin, taps, generator=g) * (2.0 / (cin * taps)) ** 0.5
        w[3] *= 2.0 ** -7                                    # a channel deep in the subnormal range of another scale
        w[5] = 0                                             # an all-zero channel
        bn = [0.75 + 0.5 * torch.rand(cout, generator=g), 0.1 * torch.randn(cout, generator=g),
              0.2 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g)]
        wf, bf = unet_ref.fold_bn_exact(w, *bn)
        wq = unet_ref.quantize_e4m3(wf).to(dev)
        wd, bnd = w.to(dev), [t.to(dev) for t in bn]
        pc = nv.PackedConv(cout, cin, taps, T, dev)
        pc.pack(wq, None)
        ref = {lay: b.clone() for lay, b in pc.buf.items()}
        for b in pc.buf.values():
            b.zero_()
        qs, bias = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
        descs += pc.descs(wd, bnd, bias, 1e-5, qs)
        expect += [(f'conv taps{taps} {cout}x{cin} layout{lay}', pc.buf[lay], ref[lay]) for lay in pc.buf]
        expect.append((f'bias taps{taps}', bias, bf.to(dev)))
        keep += [wd, bnd, pc, wq, qs]
    wt = torch.randn(64, 32, 8, generator=g) * 0.2           # transposed conv: output channels on axis 1
    wtq = unet_ref.quantize_e4m3(wt, out_axis=1).to(dev)
    ref = torch.zeros(wt.numel(), dtype=T, device=dev)
    nv.call('iunet_pack_convT', code, nv.ptr(wtq), nv.ptr(ref), 64, 32, 8, nv.stream())
    dst, qs, wtd = torch.zeros(wt.numel(), dtype=T, device=dev), torch.zeros(32, device=dev), wt.to(dev)
    descs.append(nv.make_desc(wtd, dst, 32, 64, 8, 3, T, qscale=qs))
    expect.append(('convT', dst, ref))
    nv.PackTable(descs, dev).run()
    torch.cuda.synchronize()
    for name, got, want in expect:
        bits = torch.int16 if got.dtype != torch.float32 else torch.int32
        assert torch.equal(got.view(bits), want.view(bits)), name


def test_logit_diff_is_exact_and_shows_nan(nv):
    """iunet_logit_diff (the calibration figure of engine_auto.py): max |a - b| and max |a| by integer atomicMax on the float bit patterns --
    exactly torch's maxima whatever the order; a NaN in either tensor surfaces as NaN (which the selection rule reads as "not x2m")."""
    g = torch.Generator().manual_seed(0)
    for n in (1, 255, 4097, 3 * 2 * 64 * 64 * 64 + 5):
        a = (torch.randn(n, generator=g) * 7).cuda()
        b = a + (torch.randn(n, generator=g) * 1e-4).cuda()
        out = torch.full((2,), -1.0, device='cuda')
        nv.call('iunet_logit_diff', nv.ptr(a), nv.ptr(b), n, nv.ptr(out), nv.stream())
        torch.cuda.synchronize()
        assert out[0].item() == (a - b).abs().max().item() and out[1].item() == a.abs().max().item()
    b[17] = float('nan')
    nv.call('iunet_logit_diff', nv.ptr(a), nv.ptr(b), n, nv.ptr(out), nv.stream())
    assert torch.isnan(out[0]).item() and not torch.isnan(out[1]).item()
    a[5] = float('inf')
    nv.call('iunet_logit_diff', nv.ptr(a), nv.ptr(b), n, nv.ptr(out), nv.stream())
    assert torch.isinf(out[1]).item()
    assert nv.lib().iunet_logit_diff(None, None, 4, None, None) < 0
