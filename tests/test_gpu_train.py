"""Parity of the native training kernels and of a whole optimisation step against torch
CPU autograd on the oracle network and the reference-pinned loss oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_ref, metrics_ref
from tests.test_gpu_kernels import blocked, unblocked


@pytest.fixture(scope='module')
def nv():
    from interactive_unet import _native
    _native.lib()
    return _native


@pytest.mark.parametrize('nd,shape,cin,cout', [(2, (16, 32), 32, 32), (2, (40, 72), 64, 32), (2, (16, 32), 32, 96),
                                               (2, (40, 72), 32, 64), (2, (70, 50), 64, 128), (2, (130, 200), 32, 32),      # 2-D dy-reuse form: 64 x 32 blocks on 8-row tiles / 32 x 32 blocks on 16-row tiles (two slab rows), ragged grids, several tiles per workgroup
                                               (3, (2, 8, 16), 32, 32), (3, (6, 12, 20), 64, 64),
                                               (3, (5, 9, 17), 32, 32), (3, (3, 5, 7), 64, 32), (3, (8, 24, 48), 32, 64),      # odd extents: tiles cut on every side; many tiles per workgroup run (z-plane ring, new columns)
                                               (3, (4, 8, 16), 256, 512), (2, (16, 32), 512, 256)])      # >= 128 filter blocks: LDS-transposing slab reduce
def test_conv3_wgrad_exact_integers(nv, nd, shape, cin, cout):
    """dW = sum dy (x) shifted x through the transposing LDS reads; small integers make every
    product and partial sum exact, so the result must equal the fp32 reference bit for bit."""
    g = torch.Generator().manual_seed(11)
    N = 2
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float()
    dy = torch.randint(-1, 2, (N, cout) + shape, generator=g).float()
    conv = F.conv2d if nd == 2 else F.conv3d
    w = torch.zeros((cout, cin) + (3,) * nd, requires_grad=True)
    (conv(x, w, padding=1) * dy).sum().backward()
    ref = w.grad
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    taps = 3 ** nd
    for dt in (torch.float16, torch.bfloat16):
        xb, dyb = blocked(x, dt).cuda(), blocked(dy, dt).cuda()
        nfl = nv.lib().iunet_conv3_wgrad_slab_floats(nd, N, D, H, W, cin, cout)
        slab = torch.empty(nfl, device='cuda')
        dW = torch.full((cout, cin, taps), float('nan'), device='cuda')
        nv.call('iunet_conv3_wgrad', nv.DTYPE_CODE[dt], nd, nv.ptr(xb), cin * vox, nv.ptr(dyb), cout * vox, nv.ptr(slab),
                nv.ptr(dW), 1.0, N, D, H, W, cin, cout, nv.stream())
        torch.cuda.synchronize()
        assert torch.equal(dW.cpu().reshape(ref.shape), ref), (dt, (dW.cpu().reshape(ref.shape) - ref).abs().max())


@pytest.mark.parametrize('nd', [2, 3])
def test_bn_relu_fwd_bwd(nv, nd):
    g = torch.Generator().manual_seed(12)
    shape = (24, 40) if nd == 2 else (6, 8, 20)
    N, C = 2, 32
    y = (torch.randn((N, C) + shape, generator=g) * 1.5 + 0.3).half().float()
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = torch.randn(C, generator=g) * 0.2
    dz = torch.randn((N, C) + shape, generator=g).half().float()
    yr = y.clone().requires_grad_(True)
    ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z_ref = F.relu(F.batch_norm(yr, None, None, ga, be, training=True, eps=1e-5))
    z_ref.backward(dz)
    vox = int(np.prod(shape))
    dev = 'cuda'
    yb = blocked(y, torch.float16).to(dev)
    dims = [0] + list(range(2, 2 + nd))
    slab = torch.stack([y.sum(dims), (y * y).sum(dims)], 1).reshape(1, C, 2).contiguous().to(dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    scale, shift, mean, invstd = [torch.empty(C, device=dev) for _ in range(4)]
    nv.call('iunet_bn_finalize', nv.ptr(slab), 1, C, float(N * vox), nv.ptr(gd), nv.ptr(bd), nv.ptr(rm), nv.ptr(rv),
            0.1, 1e-5, nv.ptr(scale), nv.ptr(shift), nv.ptr(mean), nv.ptr(invstd), nv.stream())
    z = torch.empty_like(yb)
    nv.call('iunet_bn_relu_fwd', 0, nv.ptr(yb), C * vox, nv.ptr(z), C * vox, nv.ptr(scale), nv.ptr(shift), C, N, vox, nv.stream())
    torch.cuda.synchronize()
    zc = unblocked(z.float().cpu(), N, C, shape)
    assert (zc - z_ref.detach()).abs().max() < 4e-3
    assert torch.allclose(rm.cpu(), 0.1 * y.mean(dims), atol=1e-5)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * y.var(dims, unbiased=True), atol=1e-4)
    # backward uses the stored (rounded) z for the ReLU mask: feed the reference the same mask
    dzb = blocked(dz, torch.float16).to(dev)
    dy = torch.empty_like(yb)
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    npart = nv.lib().iunet_bn_bwd_num_parts(N, vox)
    bslab = torch.empty(npart * C * 2, device=dev)
    coef = torch.empty(3 * C, device=dev)
    dy2 = torch.empty_like(yb)
    nv.call('iunet_bn_relu_bwd', 0, nv.ptr(dzb), C * vox, nv.ptr(z), C * vox, nv.ptr(yb), C * vox, nv.ptr(dy), C * vox,
            nv.ptr(mean), nv.ptr(invstd), nv.ptr(gd), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgamma), nv.ptr(dbeta),
            nv.ptr(bslab), nv.ptr(coef), C, N, vox, nv.stream())
    # same without z: the mask is recomputed from y -> identical result
    nv.call('iunet_bn_relu_bwd', 0, nv.ptr(dzb), C * vox, None, C * vox, nv.ptr(yb), C * vox, nv.ptr(dy2), C * vox,
            nv.ptr(mean), nv.ptr(invstd), nv.ptr(gd), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgamma), nv.ptr(dbeta),
            nv.ptr(bslab), nv.ptr(coef), C, N, vox, nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(dy, dy2)
    mask_diff = ((zc > 0) != (z_ref.detach() > 0)).float().mean().item()
    assert mask_diff < 1e-3
    dyc = unblocked(dy.float().cpu(), N, C, shape)
    assert (dyc - yr.grad).abs().max() < 1e-2 * yr.grad.abs().max()
    assert torch.allclose(dgamma.cpu(), ga.grad, rtol=2e-3, atol=2e-2)
    assert torch.allclose(dbeta.cpu(), be.grad, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize('nd', [2, 3])
def test_maxpool_bwd_matches_autograd(nv, nd):
    g = torch.Generator().manual_seed(13)
    shape = (8, 12) if nd == 2 else (4, 6, 8)
    N, C = 2, 16
    z = torch.randint(0, 4, (N, C) + shape, generator=g).float()      # many ties: first maximum must win
    zr = z.clone().requires_grad_(True)
    pooled = (F.max_pool2d if nd == 2 else F.max_pool3d)(zr, 2)
    dp = torch.randn(pooled.shape, generator=g).half().float()
    pooled.backward(dp)
    dskip = torch.randn(z.shape, generator=g).half().float()
    ref = zr.grad + dskip
    zb, dpb, dzb = blocked(z, torch.float16).cuda(), blocked(dp, torch.float16).cuda(), blocked(dskip, torch.float16).cuda()
    osp = tuple(s // 2 for s in shape)
    Do, Ho, Wo = osp if nd == 3 else (1,) + osp
    vin, vo = int(np.prod(shape)), int(np.prod(osp))
    nv.call('iunet_maxpool_bwd', 0, nd, nv.ptr(zb), C * vin, nv.ptr(dpb), C * vo, nv.ptr(dzb), C * vin, 1, C, N, Do, Ho, Wo,
            nv.stream())
    torch.cuda.synchronize()
    got = unblocked(dzb.float().cpu(), N, C, shape)
    assert (got - ref.half().float()).abs().max() <= 2e-3


@pytest.mark.parametrize('nd,big', [(2, False), (3, False), (2, True), (3, True)])
def test_convT_backward_exact_integers(nv, nd, big):
    g = torch.Generator().manual_seed(14)
    if big:      # enough 16-voxel groups for the resident-weight data-gradient kernel, ragged x extent
        shape = (24, 52) if nd == 2 else (6, 8, 40)
    else:
        shape = (6, 20) if nd == 2 else (3, 4, 16)
    N, cin, cout = 2, 64, 32
    x = torch.randint(-2, 3, (N, cin) + shape, generator=g).float().requires_grad_(True)
    w = torch.randint(-1, 2, (cin, cout) + (2,) * nd, generator=g).float().requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    up = (F.conv_transpose2d if nd == 2 else F.conv_transpose3d)(x, w, bias=b, stride=2)
    dy = torch.randint(-1, 2, up.shape, generator=g).float()
    up.backward(dy)
    D, H, W = shape if nd == 3 else (1,) + shape
    osp = tuple(2 * s for s in shape)
    vin, vout = int(np.prod(shape)), int(np.prod(osp))
    npos = 2 ** nd
    xb, dyb = blocked(x.detach(), torch.float16).cuda(), blocked(dy, torch.float16).cuda()
    wd = w.detach().cuda()
    wpk = torch.empty(w.numel(), dtype=torch.float16, device='cuda')
    nv.call('iunet_pack_convT_dgrad', 0, nv.ptr(wd), nv.ptr(wpk), cin, cout, npos, nv.stream())
    dx = torch.full((N * cin * vin,), float('nan'), dtype=torch.float16, device='cuda')
    nv.call('iunet_convT_dgrad', 0, nd, nv.ptr(dyb), cout * vout, nv.ptr(dx), cin * vin, nv.ptr(wpk), N, D, H, W, cin, cout, nv.stream())
    nb = nv.lib().iunet_convT_wgrad_blocks(nd, N, D, H, W, cin, cout)
    wslab = torch.empty(nb * cin * cout * npos, device='cuda')
    bslab = torch.empty(nb * cout, device='cuda')
    dW = torch.full((cin * cout * npos,), float('nan'), device='cuda')
    db = torch.full((cout,), float('nan'), device='cuda')
    nv.call('iunet_convT_wgrad', 0, nd, nv.ptr(xb), cin * vin, nv.ptr(dyb), cout * vout, nv.ptr(wslab), nv.ptr(bslab),
            nv.ptr(dW), nv.ptr(db), N, D, H, W, cin, cout, nv.stream())
    torch.cuda.synchronize()
    gx = unblocked(dx.float().cpu(), N, cin, shape)
    ok = x.grad.abs() <= 2048
    assert torch.equal(gx[ok], x.grad[ok])
    assert torch.equal(dW.cpu().reshape(w.shape), w.grad)
    assert torch.equal(db.cpu(), b.grad)


@pytest.mark.parametrize('nd', [2, 3])
def test_first_conv_wgrad(nv, nd):
    g = torch.Generator().manual_seed(15)
    shape = (24, 40) if nd == 2 else (6, 10, 20)
    N, cin, cout = 2, 1, 32
    xi = torch.randint(0, 256, (N, cin) + shape, generator=g, dtype=torch.uint8)
    xf = (xi.float() / 255).half().float()
    dy = torch.randn((N, cout) + shape, generator=g).half().float()
    w = torch.zeros((cout, cin) + (3,) * nd, requires_grad=True)
    ((F.conv2d if nd == 2 else F.conv3d)(xf, w, padding=1) * dy).sum().backward()
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    taps = 3 ** nd
    nb = nv.lib().iunet_first_conv_wgrad_blocks(nd, N, D, H, W)
    slab = torch.empty(nb * cout * 112, device='cuda')
    dyb = blocked(dy, torch.float16).cuda()
    xd = xi.cuda()
    dW = torch.full((cout * cin * taps,), float('nan'), device='cuda')
    nv.call('iunet_first_conv_wgrad', 0, nd, nv.ptr(xd), 2, nv.ll_array((cin * vox, vox, H * W, W, 1)), nv.ptr(dyb),
            cout * vox, nv.ptr(slab), nv.ptr(dW), N, D, H, W, cin, cout, nv.stream())
    torch.cuda.synchronize()
    assert torch.allclose(dW.cpu().reshape(w.shape), w.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('kind', metrics_ref.KINDS)
@pytest.mark.parametrize('weighted', [False, True])
def test_head_loss_fwd_bwd_vs_reference_pinned_oracle(nv, kind, weighted):
    """Loss value and d loss / d logits of the fused HIP head+softmax+loss against
    oracle/metrics_ref.py (pinned to the reference's metrics.py by the goldens), axes=[0,2,3]."""
    _head_loss_case(nv, kind, weighted, 32, 3, (20, 28))


@pytest.mark.parametrize('C0,ncls,shape', [(64, 2, (20, 28)), (64, 4, (40, 59)), (64, 3, (16, 16)), (32, 6, (48, 50)), (64, 10, (33, 31)),
                                           (32, 4, (70, 64))])
def test_head_loss_bwd_kernel_variants(nv, C0, ncls, shape):
    """Both backward kernels (dW partials in registers; in LDS for PL * ncls > 16), sizes that end inside a 256-voxel chunk and
    that span several workgroups."""
    _head_loss_case(nv, 'dice_ce', True, C0, ncls, shape)


@pytest.mark.parametrize('C0,ncls,dtype', [(32, 2, torch.bfloat16), (64, 4, torch.float16), (32, 7, torch.float16)])
def test_head_loss_act_variants_equal_the_materialised_path(nv, C0, ncls, dtype):
    """iunet_head_loss_fwd_act / _bwd_act apply relu(scale * y + shift) while loading: every output must carry the bits of the
    plain entry points run on the tensor iunet_bn_relu_fwd stores (register and LDS variants of the backward)."""
    from interactive_unet.train_engine import LOSS_KINDS
    g = torch.Generator().manual_seed(31)
    N, shape, dev = 2, (24, 44), 'cuda'
    vox = shape[0] * shape[1]
    yraw = (torch.randn((N, C0) + shape, generator=g) * 1.5).to(dtype)
    scale, shift = (0.5 + torch.rand(C0, generator=g)).to(dev), (0.4 * torch.randn(C0, generator=g)).to(dev)
    yb = blocked(yraw.float(), dtype).to(dev)
    zb = torch.empty_like(yb)
    nv.call('iunet_bn_relu_fwd', nv.DTYPE_CODE[dtype], nv.ptr(yb), C0 * vox, nv.ptr(zb), C0 * vox, nv.ptr(scale), nv.ptr(shift), C0, N, vox, nv.stream())
    w, b = (torch.randn(ncls, C0, generator=g) * 0.3).to(dev), (torch.randn(ncls, generator=g) * 0.1).to(dev)
    lab = torch.randint(0, ncls, (N,) + shape, generator=g)
    tgt = torch.stack([(lab == c) for c in range(ncls)], 1).float().to(dev).contiguous()
    wt = (0.5 + torch.rand((N, 1) + shape, generator=g)).repeat(1, ncls, 1, 1).to(dev).contiguous()
    dt = nv.DTYPE_CODE[dtype]
    res = []
    for act in (False, True):
        nparts = nv.lib().iunet_head_loss_num_parts(N, vox)
        lslab = torch.zeros(nparts * ncls * 8, device=dev)
        out4, coef = torch.zeros(4, device=dev), torch.zeros(ncls * 3, device=dev)
        dx = torch.zeros_like(yb)
        nb = nv.lib().iunet_head_loss_bwd_num_parts(N, vox, ncls, C0)
        hslab = torch.zeros(nb * ncls * (C0 + 1), device=dev)
        if act:
            nv.call('iunet_head_loss_fwd_act', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 0,
                    LOSS_KINDS['dice_ce'], nv.ptr(lslab), nv.ptr(out4), nv.ptr(coef), nv.ptr(scale), nv.ptr(shift), N, vox, nv.stream())
            nv.call('iunet_head_loss_bwd_act', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 0,
                    nv.ptr(coef), 128.0, nv.ptr(dx), C0 * vox, nv.ptr(hslab), nv.ptr(scale), nv.ptr(shift), N, vox, nv.stream())
        else:
            nv.call('iunet_head_loss_fwd', dt, nv.ptr(zb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 0,
                    LOSS_KINDS['dice_ce'], nv.ptr(lslab), nv.ptr(out4), nv.ptr(coef), N, vox, nv.stream())
            nv.call('iunet_head_loss_bwd', dt, nv.ptr(zb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 0,
                    nv.ptr(coef), 128.0, nv.ptr(dx), C0 * vox, nv.ptr(hslab), N, vox, nv.stream())
        torch.cuda.synchronize()
        res.append((out4.cpu(), coef.cpu(), dx.cpu(), hslab.cpu()))
    for a, c in zip(*res):
        assert torch.equal(a, c)
    assert res[0][2].float().abs().max() > 0


def _head_loss_case(nv, kind, weighted, C0, ncls, shape):
    from interactive_unet.train_engine import LOSS_KINDS
    g = torch.Generator().manual_seed(16)
    N = 2
    vox = shape[0] * shape[1]
    x = torch.randn((N, C0) + shape, generator=g).half().float()
    w = torch.randn(ncls, C0, generator=g) * 0.3
    b = torch.randn(ncls, generator=g) * 0.1
    lab = torch.randint(0, ncls, (N,) + shape, generator=g)
    y = torch.stack([(lab == c) for c in range(ncls)], 1).float()
    wt = None
    if weighted:
        wt = (torch.rand((N, 1) + shape, generator=g) > 0.3).float().repeat(1, ncls, 1, 1) * \
            (0.5 + torch.rand((N, 1) + shape, generator=g)).repeat(1, ncls, 1, 1)
        y = y * (wt > 0)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    logits = F.conv2d(xr, wr.view(ncls, C0, 1, 1), bias=br)
    p = torch.softmax(logits, 1)
    want = metrics_ref.loss(kind, p.detach().numpy(), y.numpy(), None if wt is None else wt.numpy(), axes=(0, 2, 3))
    gp = torch.tensor(metrics_ref.loss_grad(kind, p.detach().numpy(), y.numpy(), None if wt is None else wt.numpy(),
                                            axes=(0, 2, 3))).float()
    p.backward(gp)
    dev = 'cuda'
    xb = blocked(x, torch.float16).to(dev)
    wd, bd, yd = w.to(dev), b.to(dev), y.to(dev).contiguous()
    wtd = None if wt is None else wt.to(dev).contiguous()
    nparts = nv.lib().iunet_head_loss_num_parts(N, vox)
    lslab = torch.empty(nparts * ncls * 8, device=dev)
    out4, coef = torch.empty(4, device=dev), torch.empty(ncls * 3, device=dev)
    nv.call('iunet_head_loss_fwd', 0, nv.ptr(xb), C0 * vox, C0, nv.ptr(wd), nv.ptr(bd), ncls, nv.ptr(yd), nv.ptr(wtd), 0,
            LOSS_KINDS[kind], nv.ptr(lslab), nv.ptr(out4), nv.ptr(coef), N, vox, nv.stream())
    dx = torch.empty_like(xb)
    nparts_b = nv.lib().iunet_head_loss_bwd_num_parts(N, vox, ncls, C0)
    hslab = torch.empty(nparts_b * ncls * (C0 + 1), device=dev)
    nv.call('iunet_head_loss_bwd', 0, nv.ptr(xb), C0 * vox, C0, nv.ptr(wd), nv.ptr(bd), ncls, nv.ptr(yd), nv.ptr(wtd), 0,
            nv.ptr(coef), 1.0, nv.ptr(dx), C0 * vox, nv.ptr(hslab), N, vox, nv.stream())
    htmp = torch.empty(ncls * (C0 + 1), device=dev)
    nv.call('iunet_reduce_slab', nv.ptr(hslab), nparts_b, ncls * (C0 + 1), nv.ptr(htmp), 1.0, 0, nv.stream())
    torch.cuda.synchronize()
    o = out4.cpu()
    assert abs(o[0].item() - want) <= 2e-5 * max(1.0, abs(want)), (o[0].item(), want)
    r = metrics_ref.rounded_metrics(p.detach().numpy(), y.numpy(), None if wt is None else wt.numpy(), axes=(0, 2, 3))
    assert np.allclose(o[1:].numpy(), r, atol=2e-5)
    gx = unblocked(dx.float().cpu(), N, C0, shape)
    scale = xr.grad.abs().max().item()
    assert (gx - xr.grad).abs().max() <= 2e-3 * scale + 1e-9        # dx is stored in fp16
    ht = htmp.cpu()
    gw = ht[:ncls * C0].view(C0 // 8, ncls, 8).permute(1, 0, 2).reshape(ncls, C0)     # slab: [planes][ncls][8], then [ncls]
    assert torch.allclose(gw, wr.grad, rtol=1e-3, atol=1e-5 * max(1.0, wr.grad.abs().max().item()))
    assert torch.allclose(ht[ncls * C0:], br.grad, rtol=1e-3, atol=1e-6)


def test_adamw_matches_torch(nv):
    g = torch.Generator().manual_seed(17)
    n = 10007
    p = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3)
    pd = p.cuda()
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        gd = (gr * 8.0).cuda()              # scaled gradients, un-scaled inside the kernel
        nv.call('iunet_adamw_step', nv.ptr(pd), nv.ptr(gd), nv.ptr(m), nv.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step,
                1.0 / 8.0, None, nv.stream())
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('n', [10007, 4096, 3])
def test_adamw_on_the_device_state_matches_torch(nv, n):
    """iunet_adamw_step_dev (the training handle's optimiser step: loss scale, step count and bias corrections live in an 8-word device
    state; four parameters per thread, a scalar tail) against torch.optim.AdamW over three steps, with n not a multiple of four too."""
    g = torch.Generator().manual_seed(19)
    p = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3)
    pd = p.cuda()
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    state = torch.zeros(8, device='cuda')
    state[0] = 8.0                                                     # loss scale; words 1..3, 7 are integers (zero: step 0, fixed scale)
    for step in range(3):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        gd = (gr * 8.0).cuda()
        nv.call('iunet_adamw_step_dev', nv.ptr(pd), nv.ptr(gd), nv.ptr(m), nv.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, nv.ptr(state), 1, 1.0,
                nv.stream())
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert state.view(torch.int32)[1].item() == 3 and state.view(torch.int32)[3].item() == 0
    gd = torch.full((n,), float('inf'), device='cuda')                 # an overflowing gradient: the step is skipped and not counted
    before = pd.clone()
    nv.call('iunet_adamw_step_dev', nv.ptr(pd), nv.ptr(gd), nv.ptr(m), nv.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, nv.ptr(state), 1, 1.0, nv.stream())
    torch.cuda.synchronize()
    assert torch.equal(pd, before) and state.view(torch.int32)[1].item() == 3


@pytest.mark.parametrize('dim,shape,dtype', [(2, (64, 96), 'fp16'), (3, (16, 32, 32), 'bf16')])
def test_full_train_step_vs_autograd(dim, shape, dtype):
    """One native optimisation step (BatchNorm batch statistics, MCC+CE loss, backward, AdamW)
    against fp32 CPU autograd on the oracle network with the same weights and batch."""
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    import warnings
    torch.manual_seed(0)
    N, ncls = 2, 2
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(lr=1e-3, num_classes=ncls, dim=dim, act_dtype=dtype, pretrained=False)
    p0 = unet_ref.init_params(dim=dim, ncls=ncls, seed=5)
    model.load_named(p0)
    model = model.cuda()
    from scipy import ndimage
    rng = np.random.default_rng(0)
    img = np.stack([ndimage.gaussian_filter(rng.random(shape), 2) for _ in range(N)])
    img = (255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)[:, None]
    lab = img[:, 0] > 127
    y = np.stack([~lab, lab], 1).astype(np.float32)
    wt = np.repeat((rng.random((N, 1) + shape) > 0.2).astype(np.float32), ncls, 1)
    y = y * wt
    X = torch.tensor(img.astype(np.float32) / 255.0)
    # ---- oracle: CPU autograd, (a) pure fp32 and (b) with the HIP path's storage rounding
    # (activations, weights and activation gradients rounded to act_dtype at the same points)
    act = torch.float16 if dtype == 'fp16' else torch.bfloat16
    axes = (0,) + tuple(range(2, 2 + dim))

    def oracle_grads(act_dtype):
        pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p0.items()}
        st = {}
        logits = unet_ref.forward_logits(pr, X, dim=dim, training=True, act_dtype=act_dtype, bn_stats_out=st)
        probs = torch.softmax(logits, 1)
        lv = metrics_ref.loss('mcc_ce', probs.detach().numpy(), y, wt, axes=axes)
        gp = torch.tensor(metrics_ref.loss_grad('mcc_ce', probs.detach().numpy(), y, wt, axes=axes)).float()
        probs.backward(gp)
        return pr, st, lv
    pr32, stats, lv32 = oracle_grads(None)
    pr, _, lv = oracle_grads(act)
    # ---- native
    te = TrainEngine(model, lr=1e-3, loss_kind='mcc_ce', loss_scale=(256.0 if dtype == 'fp16' else 1.0))
    out = te.train_step(X, torch.tensor(y), torch.tensor(wt))
    torch.cuda.synchronize()
    print(f'{dim}-D {dtype}: native loss {out["Loss"]:.5f} vs oracle(same rounding) {lv:.5f} vs oracle(fp32) {lv32:.5f}')
    assert abs(out['Loss'] - lv) < (2e-3 if dtype == 'fp16' else 1e-2)
    # Gradients.  16-bit storage makes the deep-path gradients of this small random net noisy
    # (ReLU-mask / max-pool-routing flips): the same-rounding ORACLE itself is only ~0.98 (fp16) /
    # ~0.93 (bf16) cosine-close to its own fp32 gradients on enc1..dec2 (DESIGN.md "Parity").  So:
    # shallow tensors (head, dec0) must match tightly, and every tensor must be at least as close
    # to the fp32 gradients as the same-rounding oracle is (minus a small margin), with equal norms.
    cosine = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    worst_gap, report = 0.0, []
    for name in te.names:
        gn = te.g(name).cpu().reshape(pr[name].shape) / te.loss_scale
        c_native = cosine(gn, pr32[name].grad)
        c_oracle = cosine(pr[name].grad, pr32[name].grad)
        report.append((name, c_native, c_oracle))
        worst_gap = max(worst_gap, c_oracle - c_native)
        assert c_native > c_oracle - (0.02 if dtype == 'fp16' else 0.04), (name, c_native, c_oracle)
        assert c_native > 0.85, (name, c_native)
        if name.startswith('head') or name.startswith('dec0.conv') or name.startswith('dec0.bn'):
            assert c_native > (0.9995 if dtype == 'fp16' else 0.999), (name, c_native)
        nrm = (gn.norm() / (pr32[name].grad.norm() + 1e-20)).item()
        assert 0.9 < nrm < 1.1, (name, nrm)
    print(f'   min cos(native, fp32) = {min(r[1] for r in report):.4f}, min cos(same-rounding oracle, fp32) = '
          f'{min(r[2] for r in report):.4f}, worst gap = {worst_gap:.4f}')
    # running statistics updated like torch BatchNorm (momentum 0.1)
    mean, var = stats['enc0.bn1']
    rm = model.tensor('enc0.bn1.running_mean').cpu()
    assert torch.allclose(rm, 0.1 * mean, atol=2e-3)
    # parameters moved by one AdamW step: |delta| ~ lr
    d = (model.tensor('enc1.conv1.weight').cpu() - p0['enc1.conv1.weight']).abs()
    assert 0.2e-3 < d.mean().item() < 1.2e-3
    # eval forward after the step uses the updated weights / running stats
    ev = te.eval_step(X, torch.tensor(y), torch.tensor(wt))
    assert np.isfinite(ev['Loss'])


@pytest.mark.parametrize('nd,cin,big,lay', [(3, 32, False, 2), (3, 64, False, 2), (2, 32, False, 2), (2, 64, False, 2), (3, 64, True, 2),
                                            (2, 32, False, 3), (2, 64, False, 3)])       # layout 3 in 2-D: the cross-pair step
def test_conv3_fwd_and_wgrad_with_fused_input_activation(nv, nd, cin, big, lay):
    """iunet_conv3_fwd_act / iunet_conv3_wgrad_act (input = relu(scale * y + shift) applied while staging) equal
    the unfused sequence bn_relu_fwd -> conv3_fwd / conv3_wgrad bit for bit (same rounding of the activation).
    big: a grid walked in tile pairs -- the fused launch stages its halo tiles through registers, the plain one by LDS-DMA."""
    g = torch.Generator().manual_seed(21)
    T, dt = torch.bfloat16, 1
    N, cout = 2, 32
    shape = (58, 62, 120) if big else (6, 12, 20) if nd == 3 else (24, 44)
    assert nv.lib().iunet_conv3_tile_pairs(nd, N, *(shape if nd == 3 else (1,) + shape), cin, cout) == int(big)
    D, H, W = shape if nd == 3 else (1,) + shape
    taps = 3 ** nd
    vox = D * H * W
    y1 = torch.randn((N, cin) + shape, generator=g)
    scale, shift = (0.5 + torch.rand(cin, generator=g)).cuda(), (0.3 * torch.randn(cin, generator=g)).cuda()
    w = (torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.05).cuda()
    dy = torch.randn((N, cout) + shape, generator=g)
    yb, dyb = blocked(y1, T).cuda(), blocked(dy, T).cuda()
    s = nv.stream()
    z = torch.empty_like(yb)
    nv.call('iunet_bn_relu_fwd', dt, nv.ptr(yb), cin * vox, nv.ptr(z), cin * vox, nv.ptr(scale), nv.ptr(shift), cin, N, vox, s)
    pm = 6 if lay == 3 else 2
    if lay == 3 and os.environ.get('IUNET_NO_COMPACT2D'):
        pytest.skip('A/B switch IUNET_NO_COMPACT2D: no layout 3 in 2-D')
    assert lay == 2 or nv.lib().iunet_conv3_compact_ok(nd, N, D, H, W, cin, cout, 1, 0) == 1
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pm), dtype=T, device='cuda')
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pm, s)
    nt = nv.lib().iunet_conv3_stats_parts(nd, N, D, H, W, cout, lay)
    outs, stats = [], []
    for fused in (False, True):
        out = torch.full((N * cout * vox,), float('nan'), dtype=T, device='cuda')
        st = torch.zeros(nt * cout * 2, device='cuda')
        if fused:
            nv.call('iunet_conv3_fwd_act', dt, nd, nv.ptr(yb), cin * vox, nv.ptr(out), cout * vox, nv.ptr(wpk), None,
                    nv.ptr(st), nv.ptr(scale), nv.ptr(shift), N, D, H, W, cin, cout, 0, lay, s)
        else:
            nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(z), cin * vox, nv.ptr(out), cout * vox, nv.ptr(wpk), None,
                    nv.ptr(st), N, D, H, W, cin, cout, 0, lay, s)
        outs.append(out); stats.append(st)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(stats[0], stats[1])
    assert (z.float() == 0).float().mean() > 0.1          # the ReLU does clip a good part of the input
    nfl = nv.lib().iunet_conv3_wgrad_slab_floats(nd, N, D, H, W, cin, cout)
    slab = torch.empty(nfl, device='cuda')
    dws = []
    for fused in (False, True):
        dW = torch.full((cout, cin, taps), float('nan'), device='cuda')
        if fused:
            nv.call('iunet_conv3_wgrad_act', dt, nd, nv.ptr(yb), cin * vox, nv.ptr(dyb), cout * vox, nv.ptr(slab), nv.ptr(dW),
                    1.0, nv.ptr(scale), nv.ptr(shift), N, D, H, W, cin, cout, s)
        else:
            nv.call('iunet_conv3_wgrad', dt, nd, nv.ptr(z), cin * vox, nv.ptr(dyb), cout * vox, nv.ptr(slab), nv.ptr(dW),
                    1.0, N, D, H, W, cin, cout, s)
        dws.append(dW)
    torch.cuda.synchronize()
    assert torch.equal(dws[0], dws[1])


@pytest.mark.parametrize('nd,cin,lay', [(3, 32, 2), (3, 64, 2), (2, 32, 2), (2, 64, 2), (2, 32, 3), (2, 64, 3)])
def test_dgrad_with_fused_batchnorm_backward_sums(nv, nd, cin, lay):
    """iunet_conv3_dgrad_bnstats_lay: the data-gradient launch whose epilogue also accumulates the BatchNorm-backward sums of the layer its
    output flows into (sum dz', sum dz' xhat with dz' = dz where relu(bn(yp)) > 0, on the STORED dz) -- dz equals the plain launch on the
    same layout bit for bit, the sums equal a float64 reduction of the stored tensors.  Layout 3 in 2-D = the cross-pair step with
    resident weights (the training step's level-0 / level-1 data gradients)."""
    g = torch.Generator().manual_seed(33)
    T, dt = torch.bfloat16, 1
    N, cout = 2, 32
    shape = (6, 12, 20) if nd == 3 else (40, 72)
    D, H, W = shape if nd == 3 else (1,) + shape
    taps, vox = 3 ** nd, D * H * W
    if lay == 3 and (os.environ.get('IUNET_NO_COMPACT2D') or os.environ.get('IUNET_NO_COMPACT2D_BW')):
        pytest.skip('A/B switch: no fused sums on layout 3')
    assert nv.lib().iunet_conv3_compact_ok(nd, N, D, H, W, cin, cout, 0, 1) == int(nd == 2)
    assert nv.lib().iunet_conv3_compact_ok(2, N, 1, H, W, 128, cout, 0, 1) == 0          # beyond 64 channels: the plain compact launch
    dy = torch.randn((N, cin) + shape, generator=g)
    yp = torch.randn((N, cout) + shape, generator=g)
    w = (torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.05).cuda()
    mean, invstd = (0.2 * torch.randn(cout, generator=g)).cuda(), (0.5 + torch.rand(cout, generator=g)).cuda()
    scale, shift = (0.5 + torch.rand(cout, generator=g)).cuda(), (0.3 * torch.randn(cout, generator=g)).cuda()
    dyb, ypb = blocked(dy, T).cuda(), blocked(yp, T).cuda()
    s = nv.stream()
    pm = 6 if lay == 3 else 2
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pm), dtype=T, device='cuda')
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pm, s)
    nt = nv.lib().iunet_conv3_stats_parts(nd, N, D, H, W, cout, 2)
    dz = [torch.full((N * cout * vox,), float('nan'), dtype=T, device='cuda') for _ in range(2)]
    st = torch.full((nt * cout * 2,), float('nan'), device='cuda')
    nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(dyb), cin * vox, nv.ptr(dz[0]), cout * vox, nv.ptr(wpk), None, None, N, D, H, W, cin, cout, 0, lay, s)
    nv.call('iunet_conv3_dgrad_bnstats_lay', dt, nd, nv.ptr(dyb), cin * vox, nv.ptr(dz[1]), cout * vox, nv.ptr(wpk), nv.ptr(st),
            nv.ptr(ypb), cout * vox, nv.ptr(mean), nv.ptr(invstd), nv.ptr(scale), nv.ptr(shift), N, D, H, W, cin, cout, lay, s)
    torch.cuda.synchronize()
    assert torch.equal(dz[0].view(torch.int16), dz[1].view(torch.int16))
    got = st.view(nt, cout, 2).double().sum(0).cpu()
    dzf = unblocked(dz[1].cpu(), N, cout, shape).double()
    ypf = yp.to(T).double()
    bc = lambda v: v.cpu().double().view(1, cout, *([1] * nd))
    zz = (bc(scale) * ypf + bc(shift)).float().to(T).float()                       # the stored activation of the producer layer
    d = torch.where(zz > 0, dzf, torch.zeros_like(dzf))
    red = tuple(i for i in range(nd + 2) if i != 1)
    want = torch.stack([d.sum(red), (d * (ypf - bc(mean)) * bc(invstd)).sum(red)], 1)
    size = torch.stack([d.abs().sum(red), (d * (ypf - bc(mean)) * bc(invstd)).abs().sum(red)], 1)
    assert ((got - want).abs() <= 2e-5 * size + 1e-6).all(), ((got - want).abs() / size).max().item()
    assert (zz > 0).double().mean() > 0.2 and (zz > 0).double().mean() < 0.8


@pytest.mark.parametrize('nd,ch,lay,shape', [(3, 32, 2, (64, 64, 64)), (3, 64, 2, (32, 64, 64)), (2, 32, 3, (256, 512)), (2, 64, 3, (256, 256))])
def test_dgrad_with_per_sample_fused_sums(nv, nd, ch, lay, shape):
    """iunet_conv3_dgrad_sample_bnstats (GroupNorm training: the data gradient whose epilogue accumulates the backward sums of the layer its
    output flows into, with that layer's PER-SAMPLE mean / invstd / scale / shift rows, the samples along the grid's z): dz equals the plain
    launch bit for bit, the rows [N][rows][C][2] sum to each sample's (sum dz', sum dz' xhat) of the stored tensors."""
    g = torch.Generator().manual_seed(37)
    T, dt = torch.bfloat16, 1
    N = 2
    D, H, W = shape if nd == 3 else (1,) + shape
    taps, vox = 3 ** nd, D * H * W
    rows = nv.lib().iunet_conv3_sample_stats_rows(dt, nd, N, D, H, W, ch, ch, lay)
    assert rows > 0
    dy = torch.randn((N, ch) + shape, generator=g)
    yp = torch.randn((N, ch) + shape, generator=g) * torch.tensor([0.8, 1.5]).view(N, 1, *([1] * nd))
    w = (torch.randn((ch, ch) + (3,) * nd, generator=g) * 0.05).cuda()
    par = [(0.2 * torch.randn(N, ch, generator=g)).cuda(), (0.5 + torch.rand(N, ch, generator=g)).cuda(),
           (0.5 + torch.rand(N, ch, generator=g)).cuda(), (0.3 * torch.randn(N, ch, generator=g)).cuda()]      # mean, invstd, scale, shift rows
    dyb, ypb = blocked(dy, T).cuda(), blocked(yp, T).cuda()
    s = nv.stream()
    pm = 6 if lay == 3 else 2
    wpk = torch.empty(nv.pack_conv3_elems(ch, ch, taps, pm), dtype=T, device='cuda')
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), ch, ch, taps, pm, s)
    dz = [torch.full((N * ch * vox,), float('nan'), dtype=T, device='cuda') for _ in range(2)]
    st = torch.full((N * rows * ch * 2,), float('nan'), device='cuda')
    nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(dyb), ch * vox, nv.ptr(dz[0]), ch * vox, nv.ptr(wpk), None, None, N, D, H, W, ch, ch, 0, lay, s)
    nv.call('iunet_conv3_dgrad_sample_bnstats', dt, nd, nv.ptr(dyb), ch * vox, nv.ptr(dz[1]), ch * vox, nv.ptr(wpk), nv.ptr(st), nv.ptr(ypb), ch * vox,
            *[nv.ptr(t) for t in par], N, D, H, W, ch, ch, lay, s)
    torch.cuda.synchronize()
    assert torch.equal(dz[0].view(torch.int16), dz[1].view(torch.int16))
    got = st.view(N, rows, ch, 2).double().sum(1).cpu()
    dzf = unblocked(dz[1].cpu(), N, ch, shape).double()
    ypf = yp.to(T).double()
    bc = lambda v: v.cpu().double().view(N, ch, *([1] * nd))
    mean, invstd, scale, shift = [bc(t) for t in par]
    zz = (scale * ypf + shift).float().to(T).float()
    d = torch.where(zz > 0, dzf, torch.zeros_like(dzf))
    red = tuple(range(2, nd + 2))
    want = torch.stack([d.sum(red), (d * (ypf - mean) * invstd).sum(red)], 2)
    size = torch.stack([d.abs().sum(red), (d * (ypf - mean) * invstd).abs().sum(red)], 2)
    assert ((got - want).abs() <= 2e-5 * size + 1e-6).all(), ((got - want).abs() / size).max().item()


@pytest.mark.parametrize('nd,cin,cout,lay,shape', [(3, 32, 32, 2, (64, 64, 64)), (3, 64, 32, 3, (64, 64, 64)), (3, 64, 64, 3, (32, 64, 64)),
                                                    (2, 32, 32, 3, (256, 512)), (2, 128, 64, 3, (256, 256))])       # >= 8 bricks per sample
def test_conv_with_per_sample_statistics(nv, nd, cin, cout, lay, shape):
    """iunet_conv3_fwd_sample_stats (GroupNorm training: the conv's statistics epilogue per SAMPLE, the samples along the grid's z): the
    output equals iunet_conv3_fwd bit for bit, the rows [N][rows][Cout][2] sum to each sample's (sum, sum of squares), and
    iunet_gn_relu_fwd_rows on them equals iunet_gn_relu_fwd's own statistics pass within the rounding of the stored tensor."""
    g = torch.Generator().manual_seed(35)
    T, dt = torch.bfloat16, 1
    N = 3
    D, H, W = shape if nd == 3 else (1,) + shape
    taps, vox = 3 ** nd, D * H * W
    rows = nv.lib().iunet_conv3_sample_stats_rows(dt, nd, N, D, H, W, cin, cout, lay)
    assert rows > 0 and nv.lib().iunet_conv3_sample_stats_rows(dt, nd, N, D, H, W, cin, cout, 1) == 0
    tiny = (4, 8, 16) if nd == 3 else (16, 32)
    assert nv.lib().iunet_conv3_sample_stats_rows(dt, nd, N, *((tiny if nd == 3 else (1,) + tiny)), cin, cout, lay) == 0      # one brick per sample: the caller's own pass
    x = torch.randn((N, cin) + shape, generator=g) * torch.tensor([0.5, 1.0, 2.0]).view(N, 1, *([1] * nd)) + torch.tensor([0.0, 0.3, -0.2]).view(N, 1, *([1] * nd))
    w = (torch.randn((cout, cin) + (3,) * nd, generator=g) * 0.05).cuda()
    xb = blocked(x, T).cuda()
    s = nv.stream()
    pm = 6 if lay == 3 else 2
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pm), dtype=T, device='cuda')
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pm, s)
    y = [torch.full((N * cout * vox,), float('nan'), dtype=T, device='cuda') for _ in range(2)]
    st = torch.full((N * rows * cout * 2,), float('nan'), device='cuda')
    nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(xb), cin * vox, nv.ptr(y[0]), cout * vox, nv.ptr(wpk), None, None, N, D, H, W, cin, cout, 0, lay, s)
    nv.call('iunet_conv3_fwd_sample_stats', dt, nd, nv.ptr(xb), cin * vox, nv.ptr(y[1]), cout * vox, nv.ptr(wpk), nv.ptr(st), N, D, H, W, cin, cout, lay, s)
    torch.cuda.synchronize()
    assert torch.equal(y[0].view(torch.int16), y[1].view(torch.int16))
    got = st.view(N, rows, cout, 2).double().sum(1).cpu()
    yf = unblocked(y[1].cpu(), N, cout, shape).double()
    red = tuple(range(2, nd + 2))
    want = torch.stack([yf.sum(red), (yf * yf).sum(red)], 2)
    size = torch.stack([yf.abs().sum(red), (yf * yf).sum(red)], 2)
    assert ((got - want).abs() <= 1e-3 * size).all(), ((got - want).abs() / size).max().item()
    gamma, beta = (0.5 + torch.rand(cout, generator=g)).cuda(), (0.2 * torch.randn(cout, generator=g)).cuda()
    outs = []
    for r in (0, rows):
        z = torch.empty_like(y[1])
        par = [torch.empty(N * cout, device='cuda') for _ in range(4)]
        slab = st if r else torch.empty(nv.lib().iunet_gn_num_parts(N, vox) * cout * 2, device='cuda')
        nv.call('iunet_gn_relu_fwd_rows', dt, nv.ptr(y[1]), cout * vox, nv.ptr(z), cout * vox, nv.ptr(gamma), nv.ptr(beta), 8, 1e-5, nv.ptr(slab), r,
                *[nv.ptr(t) for t in par], cout, N, vox, s)
        outs.append((z, par))
    torch.cuda.synchronize()
    for a, b in zip(outs[0][1], outs[1][1]):
        assert (a - b).abs().max().item() <= 2e-3 * max(1.0, b.abs().max().item())
    assert (outs[0][0].float() - outs[1][0].float()).abs().max().item() <= 0.05      # a few bf16 ulps of an O(1) activation


@pytest.mark.parametrize('ncls,T,weighted,C0', [(2, torch.bfloat16, True, 32), (3, torch.float16, False, 32), (4, torch.bfloat16, True, 32),
                                                (4, torch.bfloat16, True, 64), (2, torch.float16, False, 64)])
def test_head_and_batchnorm_backward_in_two_passes(nv, ncls, T, weighted, C0):
    """iunet_head_bn_bwd (the head's backward + the last conv's BatchNorm + ReLU backward in two passes over that conv's raw output, the
    head's input gradient never written) against the sequence it replaces: iunet_head_loss_bwd_act -> iunet_bn_relu_bwd.  The fused
    kernel adds the logit's 32 terms plane by plane, so a gradient may sit one rounding of T away: dy within 2 ulp of T on (almost)
    every element, the reduced quantities (head dW / db, dgamma, dbeta) within 1e-4 of their magnitude."""
    g = torch.Generator().manual_seed(41)
    dt = nv.DTYPE_CODE[T]
    N = 2
    vox = 5000                                                        # three blocks per sample, the last one ragged
    y = torch.randn((N, C0, vox), generator=g) * 1.2
    yb = blocked(y, T).cuda()
    w = (torch.randn(ncls, C0, generator=g) * 0.3).cuda()
    b = (torch.randn(ncls, generator=g) * 0.1).cuda()
    lab = torch.randint(0, ncls, (N, vox), generator=g)
    tgt = torch.stack([(lab == c) for c in range(ncls)], 1).to(torch.float16).contiguous().cuda()
    wt = (torch.rand((N, ncls, vox), generator=g) > 0.2).to(torch.float16).contiguous().cuda() if weighted else None
    gamma = (0.5 + torch.rand(C0, generator=g)).cuda()
    mean, invstd = (0.1 * torch.randn(C0, generator=g)).cuda(), (0.6 + torch.rand(C0, generator=g)).cuda()
    scale = (gamma * invstd).contiguous()
    shift = (0.2 * torch.randn(C0, generator=g)).cuda()
    coef = torch.tensor([[-0.8e-4, 1.9e-4, 1.1e-4], [0.5e-4, -1.2e-4, 0.9e-4], [0.2e-4, 0.7e-4, 1.0e-4], [-0.3e-4, 0.4e-4, 0.6e-4]])[:ncls].contiguous().cuda()
    lscale = 1024.0
    s = nv.stream()
    parts = nv.lib().iunet_head_loss_bwd_num_parts(N, vox, ncls, C0)
    assert parts == nv.lib().iunet_bn_bwd_num_parts(N, vox) and nv.lib().iunet_head_bn_bwd_ok(C0, ncls) == 1 and nv.lib().iunet_head_bn_bwd_ok(96, ncls) == 0 and nv.lib().iunet_head_bn_bwd_ok(C0, 5) == 0
    out = []
    for fused in (False, True):
        dy = torch.full((N * C0 * vox,), float('nan'), dtype=T, device='cuda')
        hslab = torch.full((parts * ncls * (C0 + 1),), float('nan'), device='cuda')
        bnslab = torch.full((parts * C0 * 2,), float('nan'), device='cuda')
        bncoef = torch.empty(3 * C0, device='cuda')
        dgam, dbet = torch.empty(C0, device='cuda'), torch.empty(C0, device='cuda')
        dlbuf = torch.full((N * vox * ncls,), float('nan'), device='cuda')
        if fused:
            nv.call('iunet_head_bn_bwd', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), lscale, None,
                    nv.ptr(scale), nv.ptr(shift), nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(dgam), nv.ptr(dbet), nv.ptr(dy), C0 * vox,
                    nv.ptr(hslab), nv.ptr(bnslab), nv.ptr(bncoef), nv.ptr(dlbuf), N, vox, s)
        else:
            dz = torch.full((N * C0 * vox,), float('nan'), dtype=T, device='cuda')
            nv.call('iunet_head_loss_bwd_act', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), lscale,
                    nv.ptr(dz), C0 * vox, nv.ptr(hslab), nv.ptr(scale), nv.ptr(shift), N, vox, s)
            nv.call('iunet_bn_relu_bwd', dt, nv.ptr(dz), C0 * vox, None, 0, nv.ptr(yb), C0 * vox, nv.ptr(dy), C0 * vox, nv.ptr(mean), nv.ptr(invstd),
                    nv.ptr(gamma), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgam), nv.ptr(dbet), nv.ptr(bnslab), nv.ptr(bncoef), C0, N, vox, s)
        hrow = torch.empty(ncls * (C0 + 1), device='cuda')
        nv.call('iunet_reduce_slab', nv.ptr(hslab), parts, ncls * (C0 + 1), nv.ptr(hrow), 1.0, 0, s)
        torch.cuda.synchronize()
        out.append((dy.float().cpu(), hrow.cpu(), dgam.cpu(), dbet.cpu(), bncoef.cpu()))
    (dy0, h0, dg0, db0, c0), (dy1, h1, dg1, db1, c1) = out
    assert torch.isfinite(dy1).all() and dy0.abs().max() > 0
    ulp = 2.0 ** (-7 if T == torch.bfloat16 else -10)
    d = (dy0 - dy1).abs()
    tol = 2 * ulp * torch.maximum(dy0.abs(), dy1.abs()) + 1e-6 * dy0.abs().max()
    assert (d > tol).float().mean().item() < 1e-4, ((d > tol).float().mean().item(), d.max().item(), dy0.abs().max().item())
    for a, bb in ((h0, h1), (dg0, dg1), (db0, db1), (c0, c1)):
        assert (a - bb).abs().max().item() <= 1e-4 * a.abs().max().item() + 1e-7, ((a - bb).abs().max().item(), a.abs().max().item())


@pytest.mark.parametrize('nparts', [1, 37, 300, 2048, 8192, 8195])
def test_bn_finalize_over_many_rows(nv, nparts):
    """iunet_bn_finalize on slabs of 1 .. 8 195 rows (the first conv writes one row per TILE: 8 192 at 2 x 128^3; from 2 048 rows on the
    kernel runs on 1 024 threads) against float64: scale / shift / mean / invstd and the running statistics."""
    g = torch.Generator().manual_seed(43)
    C, count, eps, mom = 32, 4096.0 * nparts, 1e-5, 0.1
    s1 = torch.randn((nparts, C), generator=g) * 10 + 3.0 * 4096
    s2 = s1 * s1 / 4096 + 4096 * (0.5 + torch.rand((nparts, C), generator=g))          # sum of squares >= (sum)^2 / n per row
    slab = torch.stack([s1, s2], 2).contiguous().cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=g)), 0.2 * torch.randn(C, generator=g)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    out = [torch.empty(C, device='cuda') for _ in range(4)]
    gd, bd = gamma.cuda(), beta.cuda()                                # (kept alive over the launch)
    nv.call('iunet_bn_finalize', nv.ptr(slab), nparts, C, count, nv.ptr(gd), nv.ptr(bd), nv.ptr(rm), nv.ptr(rv), mom, eps,
            *[nv.ptr(t) for t in out], nv.stream())
    torch.cuda.synchronize()
    mean = slab[..., 0].double().sum(0).cpu() / count
    var = (slab[..., 1].double().sum(0).cpu() / count - mean * mean).clamp(min=0)
    invstd = 1.0 / torch.sqrt(var + eps)
    want = [gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd, mean, invstd]
    for got, w in zip(out, want):
        assert (got.cpu().double() - w).abs().max().item() <= 2e-6 * max(1.0, w.abs().max().item())
    assert (rm.cpu().double() - mom * mean).abs().max().item() <= 1e-6 * max(1.0, mean.abs().max().item())
    assert (rv.cpu().double() - ((1 - mom) + mom * var * count / (count - 1))).abs().max().item() <= 1e-5 * max(1.0, var.max().item())


@pytest.mark.parametrize('nd', [2, 3])
def test_bn_relu_pool_fwd_equals_two_passes(nv, nd):
    """iunet_bn_relu_pool_fwd == iunet_bn_relu_fwd followed by iunet_maxpool_fwd, bit for bit (both outputs)."""
    g = torch.Generator().manual_seed(29)
    T, dt, dev = torch.bfloat16, 1, 'cuda'
    N, C = 2, 32
    shape = (12, 40) if nd == 2 else (4, 6, 24)
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    do = (D // 2 if nd == 3 else 1, H // 2, W // 2)
    ovox = do[0] * do[1] * do[2]
    y = blocked(torch.randn((N, C) + shape, generator=g), T).to(dev)
    scale, shift = (0.5 + torch.rand(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    s = nv.stream()
    z1, p1 = torch.zeros_like(y), torch.zeros(N * C * ovox, dtype=T, device=dev)
    nv.call('iunet_bn_relu_fwd', dt, nv.ptr(y), C * vox, nv.ptr(z1), C * vox, nv.ptr(scale), nv.ptr(shift), C, N, vox, s)
    nv.call('iunet_maxpool_fwd', dt, nd, nv.ptr(z1), C * vox, nv.ptr(p1), C * ovox, C, N, do[0], do[1], do[2], s)
    z2, p2 = torch.zeros_like(y), torch.zeros_like(p1)
    nv.call('iunet_bn_relu_pool_fwd', dt, nd, nv.ptr(y), C * vox, nv.ptr(z2), C * vox, nv.ptr(p2), C * ovox,
            nv.ptr(scale), nv.ptr(shift), C, N, do[0], do[1], do[2], s)
    torch.cuda.synchronize()
    assert torch.equal(z1.view(torch.int16), z2.view(torch.int16))
    assert torch.equal(p1.view(torch.int16), p2.view(torch.int16))


@pytest.mark.parametrize('nd', [2, 3])
def test_bn_relu_pool_bwd_equals_three_kernels(nv, nd):
    """iunet_bn_relu_pool_bwd == iunet_maxpool_bwd (add_skip) -> iunet_bn_relu_bwd: dy bit for bit, dgamma / dbeta to fp32
    summation order."""
    g = torch.Generator().manual_seed(31)
    T, dt, dev = torch.bfloat16, 1, 'cuda'
    N, C = 2, 32
    shape = (12, 40) if nd == 2 else (4, 6, 24)
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    do = (D // 2 if nd == 3 else 1, H // 2, W // 2)
    ovox = do[0] * do[1] * do[2]
    y = blocked(torch.randn((N, C) + shape, generator=g), T).to(dev)
    dskip = blocked(torch.randn((N, C) + shape, generator=g), T).to(dev)
    dpool = (torch.randn(N * C * ovox, generator=g)).to(T).to(dev)
    gamma = (0.5 + torch.rand(C, generator=g)).to(dev)
    mean, invstd = (0.1 * torch.randn(C, generator=g)).to(dev), (0.8 + 0.4 * torch.rand(C, generator=g)).to(dev)
    beta = (0.2 * torch.randn(C, generator=g)).to(dev)
    scale = gamma * invstd
    shift = beta - mean * scale
    s = nv.stream()
    z = torch.zeros_like(y)
    nv.call('iunet_bn_relu_fwd', dt, nv.ptr(y), C * vox, nv.ptr(z), C * vox, nv.ptr(scale), nv.ptr(shift), C, N, vox, s)
    nparts = nv.lib().iunet_bn_bwd_num_parts(N, vox)
    res = []
    for fused in (False, True):
        dy = torch.zeros_like(y)
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        slab, coef = torch.zeros(nparts * C * 2, device=dev), torch.zeros(C * 3, device=dev)
        if fused:
            nv.call('iunet_bn_relu_pool_bwd', dt, nd, nv.ptr(dskip), C * vox, nv.ptr(dpool), C * ovox, nv.ptr(y), C * vox,
                    nv.ptr(dy), C * vox, nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(scale), nv.ptr(shift),
                    nv.ptr(dgam), nv.ptr(dbet), nv.ptr(slab), nv.ptr(coef), C, N, do[0], do[1], do[2], s)
        else:
            dz = dskip.clone()
            nv.call('iunet_maxpool_bwd', dt, nd, nv.ptr(z), C * vox, nv.ptr(dpool), C * ovox, nv.ptr(dz), C * vox, 1, C, N,
                    do[0], do[1], do[2], s)
            nv.call('iunet_bn_relu_bwd', dt, nv.ptr(dz), C * vox, None, 0, nv.ptr(y), C * vox, nv.ptr(dy), C * vox,
                    nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgam), nv.ptr(dbet),
                    nv.ptr(slab), nv.ptr(coef), C, N, vox, s)
        res.append((dy, dgam, dbet))
    torch.cuda.synchronize()
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-4) and torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-4)
    d = (res[0][0].float() - res[1][0].float()).abs().max().item()
    assert d <= 2e-2 * res[0][0].float().abs().max().item() * 2 ** -7 + 1e-6, d     # identical up to the fp32 order of the sums


@pytest.mark.parametrize('nd', [2, 3])
def test_first_conv_wgrad_with_folded_bn_backward(nv, nd):
    """iunet_bn_relu_bwd(dy = NULL) + iunet_first_conv_wgrad_bn == iunet_bn_relu_bwd + iunet_first_conv_wgrad, bit for bit."""
    g = torch.Generator().manual_seed(37)
    T, dt, dev = torch.bfloat16, 1, 'cuda'
    N, cin, C = 2, 1, 32
    shape = (24, 40) if nd == 2 else (6, 10, 24)
    D, H, W = shape if nd == 3 else (1,) + shape
    vox = D * H * W
    x = torch.randint(0, 256, (N, cin) + shape, dtype=torch.uint8, generator=g).to(dev)
    y = blocked(torch.randn((N, C) + shape, generator=g), T).to(dev)
    dz = blocked(torch.randn((N, C) + shape, generator=g), T).to(dev)
    gamma = (0.5 + torch.rand(C, generator=g)).to(dev)
    mean, invstd = (0.1 * torch.randn(C, generator=g)).to(dev), (0.8 + 0.4 * torch.rand(C, generator=g)).to(dev)
    scale = gamma * invstd
    shift = (0.2 * torch.randn(C, generator=g)).to(dev) - mean * scale
    s = nv.stream()
    xs = nv.ll_array((cin * vox, vox, H * W, W, 1))
    nparts = nv.lib().iunet_bn_bwd_num_parts(N, vox)
    nb = nv.lib().iunet_first_conv_wgrad_blocks(nd, N, D, H, W)
    res = []
    for fused in (False, True):
        dy = torch.zeros_like(y)
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        slab, coef = torch.zeros(nparts * C * 2, device=dev), torch.zeros(C * 3, device=dev)
        wslab = torch.zeros(nb * C * 112, device=dev)
        dW = torch.zeros(C * cin * 3 ** nd, device=dev)
        nv.call('iunet_bn_relu_bwd', dt, nv.ptr(dz), C * vox, None, 0, nv.ptr(y), C * vox, None if fused else nv.ptr(dy), C * vox,
                nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgam), nv.ptr(dbet),
                nv.ptr(slab), nv.ptr(coef), C, N, vox, s)
        if fused:
            nv.call('iunet_first_conv_wgrad_bn', dt, nd, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], xs, nv.ptr(dz), C * vox,
                    nv.ptr(y), C * vox, nv.ptr(mean), nv.ptr(invstd), nv.ptr(coef), nv.ptr(scale), nv.ptr(shift),
                    nv.ptr(wslab), nv.ptr(dW), N, D, H, W, cin, C, s)
        else:
            nv.call('iunet_first_conv_wgrad', dt, nd, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], xs, nv.ptr(dy), C * vox,
                    nv.ptr(wslab), nv.ptr(dW), N, D, H, W, cin, C, s)
        res.append((dW, dgam, dbet))
    torch.cuda.synchronize()
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dim,shape,dtype', [(2, (64, 96), 'fp16'), (3, (16, 32, 48), 'bf16')])
def test_train_step_is_deterministic_bit_for_bit(dim, shape, dtype):
    """Race screen (SURVEY.md section 5 "repeated-run race screens"): the same training step from the same state, run
    three times, must leave bit-identical gradients, parameters, BatchNorm statistics and loss.  Every reduction on the
    path (BatchNorm statistics, weight gradients, loss sums) is a fixed-order slab reduction, no float atomics -- a race
    between workgroups or a missing barrier shows up here as a differing bit."""
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    import warnings
    N, ncls = 2, 3
    rng = np.random.default_rng(11)
    img = rng.integers(1, 256, (N, 1) + shape, dtype=np.uint8)
    lab = img[:, 0] // 86
    y = torch.tensor(np.stack([(lab == c) for c in range(ncls)], 1).astype(np.float32))
    w = torch.tensor(np.repeat((rng.random((N, 1) + shape) > 0.2).astype(np.float32), ncls, 1))
    X = torch.tensor(img)
    p0 = unet_ref.init_params(dim=dim, ncls=ncls, seed=9, randomize_bn=True)
    runs = []
    for rep in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = UNet(lr=1e-3, num_classes=ncls, dim=dim, act_dtype=dtype, pretrained=False)
        model.load_named(p0)
        model = model.cuda()
        te = TrainEngine(model, lr=1e-3, loss_kind='dice_ce')
        out = [te.train_step(X, y * w, w), te.train_step(X, y * w, w)]          # two steps: the second runs on updated weights
        torch.cuda.synchronize()
        runs.append((te.grad.clone(), te.flat.clone(), [model.tensor(n).clone() for n in model._names if 'running' in n], out))
    g0, f0, b0, o0 = runs[0]
    assert torch.isfinite(g0).all() and g0.abs().max() > 0
    for g, f, b, o in runs[1:]:
        assert torch.equal(g, g0), f'{(g != g0).sum().item()} gradient elements differ between two identical runs'
        assert torch.equal(f, f0)
        assert all(torch.equal(x, y_) for x, y_ in zip(b, b0))
        assert o == o0


@pytest.mark.parametrize('dim,shape,dtype', [(2, (128, 160), torch.float16), (3, (32, 32, 48), torch.bfloat16)])
def test_forward_is_deterministic_bit_for_bit(dim, shape, dtype):
    """The same forward three times (persistent wave-specialised convolutions, XCD-aware tile walk): identical bits."""
    from interactive_unet.engine import Engine
    p = unet_ref.init_params(dim=dim, ncls=2, seed=2, randomize_bn=True)
    e = Engine(dim=dim, ncls=2, act_dtype=dtype)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    rng = np.random.default_rng(3)
    N = 2
    x = torch.tensor(rng.integers(1, 256, (N, 1) + shape, dtype=np.uint8)).cuda()
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    outs = []
    for _ in range(3):
        lg = torch.empty((N, 2) + shape, device='cuda')
        e.infer(x, (vox, vox, H * W, W, 1), N, D, H, W, logits=lg)
        torch.cuda.synchronize()
        outs.append(lg)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
