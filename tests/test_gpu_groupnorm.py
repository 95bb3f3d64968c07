"""GroupNorm(8) variant of the canonical stage (north_star: "Conv2d/Conv3d + GroupNorm/BN + ReLU"): statistics per
(sample, group), identical at training and inference.  Kernel level against torch's F.group_norm + autograd, network level
against the oracle (oracle/unet_ref.forward_logits(norm='group')), one training step against CPU autograd."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import metrics_ref, unet_ref
from tests.test_gpu_kernels import blocked, unblocked, nv      # noqa: F401


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,C,groups,sp', [(2, 32, 8, (5, 7, 9)), (3, 64, 8, (40, 33)), (1, 256, 8, (4, 4, 4)), (2, 32, 4, (9000,))])
def test_gn_relu_fwd_bwd_vs_torch(nv, dtype, N, C, groups, sp):
    g = torch.Generator().manual_seed(0)
    y = (torch.randn((N, C) + sp, generator=g) * 1.5 + 0.3).to(dtype).float()
    y[:, 3] *= 4.0                                                   # channels of one group with different ranges
    dz = (torch.randn((N, C) + sp, generator=g)).to(dtype).float()
    gamma = 0.5 + torch.rand(C, generator=g)
    gamma[1] = 0.0                                                   # a channel with gamma = 0 still gets -invstd (m1 + xhat m2) through the group statistics
    beta = 0.3 * torch.randn(C, generator=g)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z_ref = F.relu(F.group_norm(yr, groups, gr, br, eps=1e-5))
    z_ref.backward(dz)
    vox = int(np.prod(sp))
    dev = 'cuda'
    yb, dzb = blocked(y, dtype).to(dev), blocked(dz, dtype).to(dev)
    z = torch.full_like(yb, float('nan'))
    dy = torch.full_like(yb, float('nan'))
    parts = nv.lib().iunet_gn_num_parts(N, vox)
    slab = torch.empty(parts * C * 2, device=dev)
    st = [torch.empty(N * C, device=dev) for _ in range(4)]
    coef = torch.empty(N * C * 3, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    dt = nv.DTYPE_CODE[dtype]
    nv.call('iunet_gn_relu_fwd', dt, nv.ptr(yb), C * vox, nv.ptr(z), C * vox, nv.ptr(gd), nv.ptr(bd), groups, 1e-5, nv.ptr(slab),
            nv.ptr(st[0]), nv.ptr(st[1]), nv.ptr(st[2]), nv.ptr(st[3]), C, N, vox, nv.stream())
    nv.call('iunet_gn_relu_bwd', dt, nv.ptr(dzb), C * vox, nv.ptr(yb), C * vox, nv.ptr(dy), C * vox, nv.ptr(gd), groups,
            nv.ptr(st[0]), nv.ptr(st[1]), nv.ptr(st[2]), nv.ptr(st[3]), nv.ptr(dg), nv.ptr(db), nv.ptr(slab), nv.ptr(coef), C, N, vox,
            nv.stream())
    torch.cuda.synchronize()
    ulp = 2 ** -10 if dtype == torch.float16 else 2 ** -7
    got_z = unblocked(z.float().cpu(), N, C, sp)
    got_dy = unblocked(dy.float().cpu(), N, C, sp)
    zr = z_ref.detach()
    assert (got_z - zr).abs().max() <= 1.5 * ulp * max(1.0, zr.abs().max().item())
    # mean / invstd of every (sample, group), broadcast to its channels
    cpg = C // groups
    yg = y.reshape(N, groups, -1)
    mean = yg.mean(-1)
    invstd = 1.0 / torch.sqrt(yg.var(-1, unbiased=False) + 1e-5)
    assert torch.allclose(st[2].cpu().reshape(N, C), mean.repeat_interleave(cpg, 1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(st[3].cpu().reshape(N, C), invstd.repeat_interleave(cpg, 1), rtol=1e-5, atol=1e-6)
    # the ReLU mask of an element whose z is within rounding of 0 may differ: compare gradients away from those
    sure = zr.abs() > 4 * ulp * zr.abs().max()
    dyr = yr.grad
    assert ((got_dy - dyr).abs()[sure | (zr == 0)]).max() <= 3 * ulp * max(1.0, dyr.abs().max().item()) + 4e-3 * dyr.abs().max().item()
    scale_g = max(1.0, gr.grad.abs().max().item())
    assert (dg.cpu() - gr.grad).abs().max() <= 2e-2 * scale_g, (dg.cpu() - gr.grad).abs().max()
    assert (db.cpu() - br.grad).abs().max() <= 2e-2 * max(1.0, br.grad.abs().max().item())


def _model(dim, ncls, dtype, seed=2):
    from interactive_unet.unet import UNet
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(lr=1e-3, num_classes=ncls, dim=dim, act_dtype=dtype, pretrained=False, norm='group', groups=8)
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=seed, randomize_bn=True)
    m.load_named(p)
    return m.cuda(), p


@pytest.mark.parametrize('dim,shape,dtype', [(2, (64, 96), 'fp16'), (3, (16, 32, 48), 'bf16')])
def test_groupnorm_network_forward_vs_oracle(dim, shape, dtype):
    act = torch.float16 if dtype == 'fp16' else torch.bfloat16
    m, p = _model(dim, 3, dtype)
    m.eval()
    rng = np.random.default_rng(4)
    x = torch.tensor(rng.integers(0, 256, (2, 1) + shape, dtype=np.uint8))
    got = m(x.cuda()).cpu()
    xf = x.float() / 255.0
    want = unet_ref.forward(p, xf, dim=dim, act_dtype=act, norm='group', groups=8)
    want32 = unet_ref.forward(p, xf, dim=dim, norm='group', groups=8)
    err, err32 = (got - want).abs().max().item(), (got - want32).abs().max().item()
    print(f'GroupNorm net {dim}-D {dtype}: max |prob - same-rounding oracle| = {err:.2e}, vs fp32 oracle = {err32:.2e}')
    assert err <= (3e-3 if dtype == 'fp16' else 2.4e-2)
    assert m.hparams['norm'] == 'group'
    # the BatchNorm running statistics play no role: changing them changes nothing
    with torch.no_grad():
        for n in m._names:
            if 'running' in n:
                m.tensor(n).add_(1.0)
    assert torch.equal(m(x.cuda()).cpu(), got)


@pytest.mark.parametrize('dim,shape,dtype', [(2, (64, 64), 'fp16'), (3, (16, 16, 32), 'bf16')])
def test_groupnorm_train_step_vs_autograd(dim, shape, dtype):
    from interactive_unet.train_engine import TrainEngine
    act = torch.float16 if dtype == 'fp16' else torch.bfloat16
    N, ncls = 2, 2
    m, p0 = _model(dim, ncls, dtype, seed=5)
    m.train()
    rng = np.random.default_rng(1)
    img = rng.integers(1, 256, (N, 1) + shape, dtype=np.uint8)
    lab = img[:, 0] > 127
    y = np.stack([~lab, lab], 1).astype(np.float32)
    wt = np.repeat((rng.random((N, 1) + shape) > 0.2).astype(np.float32), ncls, 1)
    y = y * wt
    X = torch.tensor(img.astype(np.float32) / 255.0)
    axes = (0,) + tuple(range(2, 2 + dim))
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p0.items()}
    logits = unet_ref.forward_logits(pr, X, dim=dim, training=True, act_dtype=act, norm='group', groups=8)
    probs = torch.softmax(logits, 1)
    lv = metrics_ref.loss('dice_ce', probs.detach().numpy(), y, wt, axes=axes)
    gp = torch.tensor(metrics_ref.loss_grad('dice_ce', probs.detach().numpy(), y, wt, axes=axes)).float()
    probs.backward(gp)
    te = TrainEngine(m, lr=1e-3, loss_kind='dice_ce')
    out = te.train_step(torch.tensor(img), torch.tensor(y), torch.tensor(wt))
    assert abs(out['Loss'] - lv) <= (2e-3 if dtype == 'fp16' else 1e-2), (out['Loss'], lv)
    scale = te.loss_scale
    worst = 1.0
    for name in ('head.weight', 'dec0.conv2.weight', 'dec0.bn2.weight', 'dec0.bn2.bias', 'dec0.conv1.weight', 'enc0.conv2.weight',
                 'enc0.bn1.weight', 'enc0.conv1.weight', 'dec1.up.weight', 'enc1.conv1.weight'):
        gn = te.g(name).cpu().reshape(-1) / scale
        go = pr[name].grad.reshape(-1)
        cos = float((gn * go).sum() / (gn.norm() * go.norm() + 1e-30))
        ratio = float(gn.norm() / (go.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > (0.98 if dtype == 'fp16' else 0.95) and 0.9 < ratio < 1.1, (name, cos, ratio)
    print(f'GroupNorm train step {dim}-D {dtype}: loss {out["Loss"]:.5f} vs oracle {lv:.5f}; worst gradient cosine {worst:.4f}')
    losses = [te.train_step(torch.tensor(img), torch.tensor(y), torch.tensor(wt))['Loss'] for _ in range(6)]
    assert losses[-1] < out['Loss']


@pytest.mark.parametrize('dim,shape,dtype', [(2, (256, 512), 'fp16'), (3, (64, 64, 64), 'bf16')])
def test_groupnorm_training_with_conv_epilogue_statistics(monkeypatch, dim, shape, dtype):
    """On grids with >= 8 bricks per sample the GroupNorm statistics of a training forward come from the conv's epilogue, per sample
    (iunet_conv3_fwd_sample_stats; the first conv's per-tile rows) instead of a pass over the conv's output: the same training steps with
    and without (IUNET_NO_GN_CONV_STATS=1) -- Python-sequenced first step, C handle from the second -- agree in loss and gradients to the
    rounding of the stored 16-bit tensor (the epilogue sums the fp32 accumulators, the pass the rounded values)."""
    from interactive_unet.train_engine import TrainEngine
    from interactive_unet import _native as nv
    N, ncls = 2, 2
    D, H, W = shape if dim == 3 else (1,) + shape
    assert nv.lib().iunet_conv3_sample_stats_rows(nv.DTYPE_CODE[torch.float16 if dtype == 'fp16' else torch.bfloat16], dim, N, D, H, W, 32, 32, 2) > 0
    rng = np.random.default_rng(3)
    img = rng.integers(1, 256, (N, 1) + shape, dtype=np.uint8)
    img[1] = (img[1] // 3) + 100                                       # the two samples differ in mean and spread
    lab = img[:, 0] > 127
    y = np.stack([~lab, lab], 1).astype(np.float32)
    wt = np.ones_like(y)
    runs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('IUNET_NO_GN_CONV_STATS', '1')
        else:
            monkeypatch.delenv('IUNET_NO_GN_CONV_STATS', raising=False)
        m, _ = _model(dim, ncls, dtype, seed=7)
        m.train()
        te = TrainEngine(m, lr=1e-3, loss_kind='dice_ce')
        assert te.gn_conv_stats == (not off)
        first = te.train_step(torch.tensor(img), torch.tensor(y), torch.tensor(wt))['Loss']
        g1 = te.grad.clone() / te.loss_scale
        rest = [te.train_step(torch.tensor(img), torch.tensor(y), torch.tensor(wt))['Loss'] for _ in range(3)]
        runs.append((first, g1, rest))
    (l0, g0, r0), (l1, g1, r1) = runs
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    print(f'GroupNorm {dim}-D {dtype}: first loss {l0:.6f} / {l1:.6f}, gradient cosine {cos:.6f}, later losses {r0} / {r1}')
    assert abs(l0 - l1) <= 2e-3 and cos > 0.999 and abs(float(g0.norm() / g1.norm()) - 1) < 1e-2
    assert all(abs(a - b) <= 5e-3 for a, b in zip(r0, r1)) and r0[-1] < l0


@pytest.mark.parametrize('ncls,T,C0', [(2, torch.bfloat16, 32), (3, torch.float16, 32), (4, torch.bfloat16, 64)])
def test_head_and_groupnorm_backward_in_two_passes(nv, ncls, T, C0):
    """iunet_head_gn_bwd (head backward + the last conv's GroupNorm + ReLU backward in two passes over that conv's raw output, per-sample
    parameter rows) against the sequence on materialised tensors: iunet_gn_relu_fwd -> iunet_head_loss_bwd -> iunet_gn_relu_bwd."""
    g = torch.Generator().manual_seed(47)
    dt = nv.DTYPE_CODE[T]
    N, vox, groups = 3, 4500, 8
    y = torch.randn((N, C0, vox), generator=g) * torch.tensor([0.7, 1.0, 1.6]).view(N, 1, 1) + torch.tensor([0.2, -0.1, 0.4]).view(N, 1, 1)
    yb = blocked(y, T).cuda()
    w = (torch.randn(ncls, C0, generator=g) * 0.3).cuda()
    b = (torch.randn(ncls, generator=g) * 0.1).cuda()
    lab = torch.randint(0, ncls, (N, vox), generator=g)
    tgt = torch.stack([(lab == c) for c in range(ncls)], 1).to(torch.float16).contiguous().cuda()
    wt = (torch.rand((N, ncls, vox), generator=g) > 0.2).to(torch.float16).contiguous().cuda()
    gamma, beta = (0.5 + torch.rand(C0, generator=g)).cuda(), (0.2 * torch.randn(C0, generator=g)).cuda()
    coef = torch.tensor([[-0.8e-4, 1.9e-4, 1.1e-4], [0.5e-4, -1.2e-4, 0.9e-4], [0.2e-4, 0.7e-4, 1.0e-4], [-0.3e-4, 0.4e-4, 0.6e-4]])[:ncls].contiguous().cuda()
    s = nv.stream()
    parts = nv.lib().iunet_head_loss_bwd_num_parts(N, vox, ncls, C0)
    z = torch.empty_like(yb)
    par = [torch.empty(N * C0, device='cuda') for _ in range(4)]          # scale, shift, mean, invstd rows
    slab = torch.empty(nv.lib().iunet_gn_num_parts(N, vox) * C0 * 2, device='cuda')
    nv.call('iunet_gn_relu_fwd', dt, nv.ptr(yb), C0 * vox, nv.ptr(z), C0 * vox, nv.ptr(gamma), nv.ptr(beta), groups, 1e-5, nv.ptr(slab),
            *[nv.ptr(t) for t in par], C0, N, vox, s)
    out = []
    for fused in (False, True):
        dy = torch.full((N * C0 * vox,), float('nan'), dtype=T, device='cuda')
        hslab = torch.full((parts * ncls * (C0 + 1),), float('nan'), device='cuda')
        bnslab = torch.full((parts * C0 * 2,), float('nan'), device='cuda')
        bncoef = torch.empty(3 * C0 * N, device='cuda')
        dgam, dbet = torch.empty(C0, device='cuda'), torch.empty(C0, device='cuda')
        dz = torch.full((N * C0 * vox,), float('nan'), dtype=T, device='cuda')      # fused: its scratch
        if fused:
            nv.call('iunet_head_gn_bwd', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), 512.0, None,
                    *[nv.ptr(t) for t in par], nv.ptr(gamma), groups, nv.ptr(dgam), nv.ptr(dbet), nv.ptr(dy), C0 * vox, nv.ptr(hslab), nv.ptr(bnslab),
                    nv.ptr(bncoef), nv.ptr(dz), N, vox, s)
        else:
            nv.call('iunet_head_loss_bwd', dt, nv.ptr(z), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), 512.0,
                    nv.ptr(dz), C0 * vox, nv.ptr(hslab), N, vox, s)
            nv.call('iunet_gn_relu_bwd', dt, nv.ptr(dz), C0 * vox, nv.ptr(yb), C0 * vox, nv.ptr(dy), C0 * vox, nv.ptr(gamma), groups, *[nv.ptr(t) for t in par],
                    nv.ptr(dgam), nv.ptr(dbet), nv.ptr(bnslab), nv.ptr(bncoef), C0, N, vox, s)
        hrow = torch.empty(ncls * (C0 + 1), device='cuda')
        nv.call('iunet_reduce_slab', nv.ptr(hslab), parts, ncls * (C0 + 1), nv.ptr(hrow), 1.0, 0, s)
        torch.cuda.synchronize()
        out.append((dy.float().cpu(), hrow.cpu(), dgam.cpu(), dbet.cpu()))
    (dy0, h0, dg0, db0), (dy1, h1, dg1, db1) = out
    assert torch.isfinite(dy1).all() and dy0.abs().max() > 0
    ulp = 2.0 ** (-7 if T == torch.bfloat16 else -10)
    d = (dy0 - dy1).abs()
    tol = 2 * ulp * torch.maximum(dy0.abs(), dy1.abs()) + 2e-3 * dy0.abs().max()       # (dy = a d - c1 - xhat c2: the terms cancel, so absolute to the tensor's scale too)
    assert (d > tol).float().mean().item() < 1e-4, ((d > tol).float().mean().item(), d.max().item(), dy0.abs().max().item())
    for a, bb in ((h0, h1), (dg0, dg1), (db0, db1)):
        assert (a - bb).abs().max().item() <= 2e-4 * a.abs().max().item() + 1e-7, ((a - bb).abs().max().item(), a.abs().max().item())


@pytest.mark.parametrize('dtype,nd,N,C,groups,do', [(torch.bfloat16, 3, 2, 32, 8, (3, 5, 6)), (torch.float16, 2, 3, 64, 8, (1, 20, 17)), (torch.bfloat16, 3, 1, 128, 8, (2, 2, 4))])
def test_gn_pooled_forward_and_backward_equal_the_unfused_sequences(nv, dtype, nd, N, C, groups, do):
    """iunet_gn_relu_pool_fwd = iunet_gn_relu_fwd + iunet_maxpool_fwd and iunet_gn_relu_pool_bwd = iunet_maxpool_bwd (add_skip) +
    iunet_gn_relu_bwd, bit for bit: every value is rounded where the unfused sequence rounds it; the gradient of the stage output is never
    written.  The skip gradient sits in the first half of a wider buffer (the concat gradient), as in the network."""
    g = torch.Generator().manual_seed(3)
    sp = tuple(2 * d for d in do) if nd == 3 else (1, 2 * do[1], 2 * do[2])
    vox, ovox = int(np.prod(sp)), int(np.prod(do))
    dev = 'cuda'
    y = (torch.randn((N, C) + sp, generator=g) * 1.5 + 0.3).to(dtype).float()
    y[:, :, ..., ::2] = y[:, :, ..., 1::2]                          # ties inside every window: the FIRST maximum takes the pooled gradient
    gamma, beta = (0.5 + torch.rand(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    dskip = torch.randn((N, 2 * C) + sp, generator=g).to(dtype).float()
    dpool = torch.randn((N, C) + do, generator=g).to(dtype).float()
    yb, dcat, dpb = blocked(y, dtype).to(dev), blocked(dskip, dtype).to(dev), blocked(dpool, dtype).to(dev)
    dt = nv.DTYPE_CODE[dtype]
    parts = nv.lib().iunet_gn_num_parts(N, vox)
    res = []
    for fused in (False, True):
        slab = torch.zeros(parts * C * 2, device=dev)
        st = [torch.zeros(N * C, device=dev) for _ in range(4)]
        coef = torch.zeros(N * C * 3, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        z, pooled = torch.zeros_like(yb), torch.zeros(N * C * ovox, dtype=dtype, device=dev)
        dy = torch.zeros_like(yb)
        dc = dcat.clone()
        if fused:
            nv.call('iunet_gn_relu_pool_fwd', dt, nd, nv.ptr(yb), C * vox, nv.ptr(z), C * vox, nv.ptr(pooled), C * ovox, nv.ptr(gamma), nv.ptr(beta), groups, 1e-5,
                    nv.ptr(slab), *[nv.ptr(t) for t in st], C, N, do[0], do[1], do[2], nv.stream())
            nv.call('iunet_gn_relu_pool_bwd', dt, nd, nv.ptr(dc), 2 * C * vox, nv.ptr(dpb), C * ovox, nv.ptr(yb), C * vox, nv.ptr(dy), C * vox, nv.ptr(gamma), groups,
                    *[nv.ptr(t) for t in st], nv.ptr(dg), nv.ptr(db), nv.ptr(slab), nv.ptr(coef), C, N, do[0], do[1], do[2], nv.stream())
        else:
            nv.call('iunet_gn_relu_fwd', dt, nv.ptr(yb), C * vox, nv.ptr(z), C * vox, nv.ptr(gamma), nv.ptr(beta), groups, 1e-5, nv.ptr(slab),
                    *[nv.ptr(t) for t in st], C, N, vox, nv.stream())
            nv.call('iunet_maxpool_fwd', dt, nd, nv.ptr(z), C * vox, nv.ptr(pooled), C * ovox, C, N, do[0], do[1], do[2], nv.stream())
            nv.call('iunet_maxpool_bwd', dt, nd, nv.ptr(z), C * vox, nv.ptr(dpb), C * ovox, nv.ptr(dc), 2 * C * vox, 1, C, N, do[0], do[1], do[2], nv.stream())
            nv.call('iunet_gn_relu_bwd', dt, nv.ptr(dc), 2 * C * vox, nv.ptr(yb), C * vox, nv.ptr(dy), C * vox, nv.ptr(gamma), groups, *[nv.ptr(t) for t in st],
                    nv.ptr(dg), nv.ptr(db), nv.ptr(slab), nv.ptr(coef), C, N, vox, nv.stream())
        torch.cuda.synchronize()
        res.append((z, pooled, dy, *st))
        grads = (dg, db)
        res[-1] += grads
    names = ('z', 'pooled', 'dy', 'scale', 'shift', 'mean', 'invstd', 'dgamma', 'dbeta')
    for name, a, b in zip(names, *res):
        if name in ('dgamma', 'dbeta'):      # (the pooled first pass walks the voxels window by window: another summation order of the same terms)
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * a.abs().max().item()), name
        elif name == 'dy':
            d = (a.float() - b.float()).abs().max().item()
            assert d <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * a.float().abs().max().item(), (name, d)
            assert (a != b).float().mean().item() < 0.01
        else:
            assert torch.equal(a, b), name
    assert res[0][2].float().abs().max().item() > 0


# ---------------------------------------------------------------------------- tolerance-meeting prediction modes
def _gn_parity(dim, shape, N, seed, modes):
    """GroupNorm network in the fp32 mode and in split precision against the fp32 CPU oracle: the gate of tests/test_gpu_parity.py
    (logits <= 1e-3 absolute, class map equal outside the tie band, IoU on rounded probabilities)."""
    from tests.test_gpu_parity import _assert_fp32_mode, _compare, _forward, _labels, _smooth
    from interactive_unet.engine_auto import EngineAuto
    from interactive_unet.engine_f32 import EngineF32
    from interactive_unet.engine_x2 import EngineX2
    ncls = 2
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=seed, randomize_bn=True)
    img = np.stack([_smooth(shape, seed * 100 + i, sigma=6) for i in range(N)])[:, None]
    x = torch.tensor(img)
    ref = unet_ref.forward_logits(p, x.float() / 255.0, dim=dim, norm='group', groups=8)
    y_true = _labels(img, ncls)
    pc = {k: v.cuda() for k, v in p.items()}
    res = {}
    for mode in modes:
        if mode == 'fp32':
            e = EngineF32(dim=dim, ncls=ncls, norm='group', groups=8)
        elif mode == 'fp16x2':
            e = EngineX2(dim=dim, ncls=ncls, norm='group', groups=8, mixed=False)
        elif mode == 'x2m':
            e = EngineX2(dim=dim, ncls=ncls, norm='group', groups=8, mixed=True)
        else:
            e = EngineAuto(dim=dim, ncls=ncls, norm='group', groups=8)
            assert e.policy == 'auto'
        e.load_eval(pc)
        res[mode] = _compare(f'GroupNorm {mode} {dim}-D {N} x {shape}', *_forward(e, x.cuda(), dim, ncls), ref, y_true)
        _assert_fp32_mode(res[mode])
        if hasattr(e, 'saturated'):
            assert not e.saturated()
        if mode == 'default':
            d = e.describe()
            print(f'    default mode of the GroupNorm network: {d["form"]} (calibration {d["calibration_max_abs_logit_diff_x2m_vs_fp16x2"]:.2e})')
            assert (d['form'] == 'x2m') == (d['calibration_max_abs_logit_diff_x2m_vs_fp16x2'] <= d['threshold'])
        del e
        torch.cuda.empty_cache()
    return res


@pytest.mark.parametrize('dim,shape', [(2, (64, 96)), (2, (40, 72)), (3, (16, 32, 48)), (3, (8, 24, 40))])
def test_groupnorm_tolerance_modes_small_shapes(dim, shape):
    r = _gn_parity(dim, shape, 2, seed=3, modes=('fp32', 'fp16x2', 'x2m'))
    assert r['fp32']['err'] <= 1e-4 and r['fp16x2']['err'] <= 1e-4 and r['x2m']['err'] <= 5e-4


def test_groupnorm_headline_2d_512_squared():
    """BASELINE.json configs[1] shape with GroupNorm(8): 2 x 512^2 within 1e-3 of oracle/unet_ref.forward_logits(norm='group')."""
    _gn_parity(2, (512, 512), 2, seed=6, modes=('fp32', 'fp16x2', 'x2m', 'default'))


def test_groupnorm_headline_3d_128_cubed():
    """BASELINE.json configs[2] shape with GroupNorm(8): one 128^3 chunk within 1e-3 of the oracle."""
    _gn_parity(3, (128, 128, 128), 1, seed=5, modes=('fp32', 'fp16x2', 'x2m', 'default'))


def test_groupnorm_unet_module_default_predicts_within_tolerance():
    """UNet(norm='group') as a user builds it: trains in fp16, `forward()` answers from the split-precision engine (fp16x2 form)."""
    from interactive_unet.engine_auto import EngineAuto
    from tests.test_gpu_parity import _smooth
    m, p = _model(2, 2, None)
    m.eval()
    assert isinstance(m.engine('eval'), EngineAuto) and m.engine('eval').policy == 'auto'
    x = torch.tensor(_smooth((96, 64), 3))[None, None]
    got = m(x.cuda()).cpu()
    want = unet_ref.forward(p, x.float() / 255.0, dim=2, norm='group', groups=8)
    assert (got - want).abs().max().item() <= 2e-4 and m.engine('eval').form in ('x2m', 'fp16x2')
    m32, _ = _model(2, 2, 'fp32')
    m32.eval()
    assert (m32(x.cuda()).cpu() - want).abs().max().item() <= 1e-5


@pytest.mark.parametrize('N,C,groups,sp', [(2, 32, 8, (5, 7, 9)), (3, 64, 8, (40, 33)), (1, 256, 8, (4, 4, 4)), (2, 32, 4, (20001,)), (1, 64, 16, (129, 130))])
def test_precise_groupnorm_kernels_vs_float64(nv, N, C, groups, sp):
    """csrc/gn_precise.hip one level below the network: relu(group_norm(x)) of the fp32 mode's planar tensors and of the split-precision
    word pairs against torch's group_norm in float64 -- segment counts that do not divide the voxels, one and several segments per plane,
    a strided output (a half of a concat buffer), channels of one group at different scales."""
    from interactive_unet.engine_x2 import EngineX2
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, C) + sp, generator=g) * 1.5 + 0.3
    x[:, 3] *= 40.0
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    vox = int(np.prod(sp))
    dev = 'cuda'
    slab = torch.empty(nv.lib().iunet_gn_precise_slab_bytes(N, C, vox), dtype=torch.uint8, device=dev)
    sc, sh = torch.empty(N * C, device=dev), torch.empty(N * C, device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    # ---- fp32 planar, output into the second half of a [N][2C][vox] buffer
    xd = x.reshape(N, C, vox).contiguous().to(dev)
    y = torch.full((N, 2 * C, vox), float('nan'), device=dev)
    nv.call('iunet_f32_gn_relu_fwd', nv.ptr(xd), C * vox, nv.c_void_p(y.data_ptr() + 4 * C * vox), 2 * C * vox, nv.ptr(gd), nv.ptr(bd), groups, 1e-5,
            nv.ptr(slab), nv.ptr(sc), nv.ptr(sh), C, N, vox, nv.stream())
    torch.cuda.synchronize()
    want = F.relu(F.group_norm(x.double(), groups, gamma.double(), beta.double(), eps=1e-5)).reshape(N, C, vox)
    got = y[:, C:].cpu().double()
    assert torch.isnan(y[:, :C]).all()                                   # the other half of the buffer is untouched
    assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    # the per-(sample, channel) affine pair it leaves behind: scale = rstd * gamma
    xg = x.double().reshape(N, groups, -1)
    rstd = 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)
    assert torch.allclose(sc.cpu().double().reshape(N, C), rstd.repeat_interleave(C // groups, 1) * gamma.double(), rtol=1e-6)
    # ---- split precision: the words of act_scale x value in, act_scale x relu(gn) out
    e = EngineX2(dim=2, mixed=False)                                     # (its layout helpers and act_scale only)
    A = e.act_scale
    x4 = x.reshape((N, C) + (sp if len(sp) > 1 else (1,) + sp))
    xs = e.to_split(x4).to(dev)                                          # [N][C/8 hi | C/8 lo][vox][8]
    ys = torch.zeros_like(xs)
    sat = torch.zeros(1, dtype=torch.int32, device=dev)
    nv.call('iunet_x2_gn_relu_fwd', nv.ptr(xs), 2 * C * vox, C // 8, nv.ptr(ys), 2 * C * vox, C // 8, nv.ptr(gd), nv.ptr(bd), groups, 1e-5, A,
            nv.ptr(slab), nv.ptr(sc), nv.ptr(sh), C, N, vox, nv.ptr(sat), nv.stream())
    torch.cuda.synchronize()
    # reference on the values the split words actually hold (22 bits of x)
    xin = e.from_split(xs.cpu(), N, C, x4.shape[2:]).double().reshape((N, C) + sp)
    want2 = F.relu(F.group_norm(xin, groups, gamma.double(), beta.double(), eps=1e-5)).reshape(N, C, vox)
    got2 = e.from_split(ys.cpu(), N, C, x4.shape[2:]).double().reshape(N, C, vox)
    assert (got2 - want2).abs().max().item() <= 2e-6 * max(1.0, want2.abs().max().item())
    assert int(sat.item()) == 0
