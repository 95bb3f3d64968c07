"""fp32 parity form of the training step (interactive_unet/train_engine_f32.py, csrc/train_f32.hip + precise_f32.hip) against CPU
autograd: the oracle's forward in training mode (oracle/unet_ref.py, plain fp32) + the host metrics' loss + torch.autograd +
the restated AdamW.  The device step differs only by the order of its fp32 sums, so the bar is tight -- it replaces the cosine gate
that the 16-bit training path can be held to (tests/test_gpu_train.py):

  loss                         |native - oracle| <= 1e-6 (relative to max(1, |loss|))
  every activation             max |native - oracle| <= 1e-4 (forward values)
  every parameter gradient     max |native - oracle| <= 1e-4 x max |oracle| per tensor  ("<= 1e-4 relative")
                               -- with the oracle's ReLU / max-pool gradients routed through the DEVICE's masks, see below
  parameters after AdamW and the BatchNorm running statistics: within 1e-6 of the same update applied to the device's gradients

Why "through the device's masks": two correct fp32 forwards differ by ~1e-5 in the activations (summation order), and a ReLU or
max-pool turns such a difference into a flipped mask bit wherever a pre-activation lies inside that noise -- one element of a
gradient SUM changes, which at test sizes (32 x 48 pixels: 3 072 elements per channel at level 0, 48 at level 3) is 1e-3 .. 2e-2 of a
parameter gradient.  The test therefore (1) holds every forward activation to 1e-4, (2) requires every flipped mask element to be
such a tie (the oracle's pre-activation smaller than the measured forward difference), and (3) holds the gradients to 1e-4 once both
sides differentiate the same piecewise-linear branch.  The raw, unaligned comparison is printed (2e-2 at these sizes, shrinking with
1 / elements).  The kernel-level pieces (weight gradient, BatchNorm backward, max-pool backward) are checked on their own first."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import unet_ref


def _nv():
    from interactive_unet import _native as nv
    nv.lib()
    return nv


@pytest.mark.parametrize('dim,shape,ci,co,taps', [(3, (6, 10, 20), 32, 32, 27), (3, (4, 4, 16), 1, 32, 27), (2, (24, 40), 64, 32, 9),
                                                   (3, (4, 6, 10), 40, 3, 1), (2, (16, 16), 256, 64, 1)])
def test_weight_gradient_kernel(dim, shape, ci, co, taps):
    nv = _nv()
    g = torch.Generator().manual_seed(1)
    N = 2
    sp = (1,) + shape if dim == 2 else shape
    vox = int(np.prod(shape))
    x = torch.randn((N, ci) + shape, generator=g)
    dy = torch.randn((N, co) + shape, generator=g)
    xd, dyd = x.cuda().contiguous(), dy.cuda().contiguous()
    splits = nv.lib().iunet_f32_wgrad_splits(dim, N, sp[0], sp[1], sp[2], ci, co)
    slab = torch.empty(splits * co * ci * taps, device='cuda')
    out = torch.empty(co * ci * taps, device='cuda')
    nv.call('iunet_f32_wgrad', dim, nv.ptr(xd), ci * vox, nv.ptr(dyd), co * vox, nv.ptr(slab), N, sp[0], sp[1], sp[2], ci, co, taps, nv.stream())
    nv.call('iunet_reduce_slab', nv.ptr(slab), splits, co * ci * taps, nv.ptr(out), 1.0, 0, nv.stream())
    torch.cuda.synchronize()
    if taps == 1:
        want = torch.einsum('nov,niv->oi', dy.double().reshape(N, co, vox), x.double().reshape(N, ci, vox)).reshape(co, ci, 1)
    else:
        w = torch.zeros((co, ci) + (3,) * dim, dtype=torch.float64, requires_grad=True)
        y = (F.conv3d if dim == 3 else F.conv2d)(x.double(), w, padding=1)
        want = torch.autograd.grad(y, w, dy.double())[0].reshape(co, ci, taps)
    got = out.cpu().double().reshape(co, ci, taps)
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-6, err


def test_batchnorm_relu_and_pool_backward_kernels():
    nv = _nv()
    g = torch.Generator().manual_seed(2)
    N, C, shape = 2, 32, (4, 6, 10)
    vox = int(np.prod(shape))
    y = torch.randn((N, C) + shape, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = (0.5 + torch.rand(C, generator=g)).double().requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)
    z = torch.relu(F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5))
    p = F.max_pool3d(z, 2)
    dp = torch.randn(p.shape, generator=g, dtype=torch.float64)
    want_dy, want_dg, want_db = torch.autograd.grad(p, [y, gamma, beta], dp)
    f = lambda t: t.detach().float().cuda().contiguous()
    yd, gd, bd = f(y), f(gamma), f(beta)
    mean, std = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    zd = torch.empty_like(yd)
    s = nv.stream()
    nv.call('iunet_f32_bn_stats', nv.ptr(yd), C * vox, C, N, vox, 1e-5, 0.1, nv.ptr(mean), nv.ptr(std), None, None, s)
    nv.call('iunet_f32_bn_relu_fwd', nv.ptr(yd), C * vox, nv.ptr(zd), C * vox, nv.ptr(mean), nv.ptr(std), nv.ptr(gd), nv.ptr(bd), C, N, vox, s)
    assert (zd.cpu().double() - z.detach()).abs().max().item() < 2e-6
    dz = torch.empty_like(yd)
    dpd = f(dp)
    nv.call('iunet_f32_maxpool_bwd', 3, nv.ptr(zd), C * vox, nv.ptr(dpd), C * vox // 8, nv.ptr(dz), C * vox, C, N, 2, 3, 5, 0, s)
    dyd, dg, db = torch.empty_like(yd), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    nv.call('iunet_f32_bn_relu_bwd', nv.ptr(dz), C * vox, nv.ptr(yd), C * vox, nv.ptr(dyd), C * vox, nv.ptr(mean), nv.ptr(std), nv.ptr(gd),
            nv.ptr(bd), nv.ptr(dg), nv.ptr(db), C, N, vox, s)
    torch.cuda.synchronize()
    for got, want in ((dyd, want_dy), (dg, want_dg), (db, want_db)):
        assert (got.cpu().double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())


class _MaskRelu(torch.autograd.Function):
    """relu whose branch is given: forward y * mask, backward g * mask"""

    @staticmethod
    def forward(ctx, y, mask):
        ctx.save_for_backward(mask)
        return y * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def _oracle_forward(p, X, dim, masks=None, pool_idx=None, acts=None):
    """oracle/unet_ref.forward_logits(training=True) restated with injectable ReLU masks / pool arg-max indices (None: its own), and
    every activation recorded in `acts` (pre-activation and output of each BatchNorm + ReLU)."""
    conv = F.conv3d if dim == 3 else F.conv2d
    convT = F.conv_transpose3d if dim == 3 else F.conv_transpose2d
    pool = F.max_pool3d if dim == 3 else F.max_pool2d
    L = 4
    red = [0] + list(range(2, 2 + dim))
    shape = [1, -1] + [1] * dim

    def stage(prefix, t):
        for j in (1, 2):
            y = conv(t, p[f'{prefix}.conv{j}.weight'], padding=1)
            mean, var = y.mean(dim=red), y.var(dim=red, unbiased=False)
            y = (y - mean.view(shape)) / torch.sqrt(var.view(shape) + 1e-5)
            y = y * p[f'{prefix}.bn{j}.weight'].view(shape) + p[f'{prefix}.bn{j}.bias'].view(shape)
            name = f'{prefix}.{j}'
            t = F.relu(y) if masks is None else _MaskRelu.apply(y, masks[name])
            if acts is not None:
                acts[name] = (y.detach(), t.detach(), mean.detach(), var.detach())
        return t
    skips, t = [], X
    for l in range(L):
        t = stage(f'enc{l}', t)
        if l < L - 1:
            skips.append(t)
            if pool_idx is None:
                t = pool(t, 2)
            else:
                idx = pool_idx[l]
                t = torch.gather(t.flatten(2), 2, idx.flatten(2)).view(idx.shape)
    for l in range(L - 2, -1, -1):
        up = convT(t, p[f'dec{l}.up.weight'], bias=p[f'dec{l}.up.bias'], stride=2)
        t = stage(f'dec{l}', torch.cat([skips[l], up], dim=1))
    return conv(t, p['head.weight'], bias=p['head.bias'])


def _device_acts(te, ws, N, dim):
    """the device's activations by the oracle's names: (z tensor [N, C, *sp]) for every BatchNorm + ReLU output"""
    L, ch, dims = te.levels, te.ch, ws['dims']
    out = {}
    sp = lambda l: tuple(dims[l][1:]) if dim == 2 else tuple(dims[l])
    for prefix in te.stage_names():
        ci, co, l = te.stage_io(prefix)
        out[f'{prefix}.1'] = ws['z1.' + prefix].view((N, co) + sp(l)).cpu()
        if prefix.startswith('enc') and l < L - 1:
            out[f'{prefix}.2'] = ws[f'cat{l}'].view((N, 2 * co) + sp(l))[:, :co].cpu()
        else:
            out[f'{prefix}.2'] = ws[f'b{l}'].view((N, co) + sp(l)).cpu()
    return out


@pytest.mark.parametrize('dim,shape,ncls,loss_name,weighted', [
    (2, (32, 48), 2, 'mcc_ce_loss', True),
    (2, (64, 64), 3, 'dice_ce_loss', False),
    (3, (8, 16, 16), 2, 'mcc_ce_loss', True),
    (3, (16, 8, 24), 4, 'iou_loss', True),
])
def test_whole_step_against_cpu_autograd(dim, shape, ncls, loss_name, weighted):
    from interactive_unet.unet import UNet
    from interactive_unet import metrics
    from interactive_unet.train_engine_f32 import TrainEngineF32
    N, lr = 2, 1e-3
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=13, randomize_bn=True)
    rng = np.random.default_rng(5)
    X = torch.tensor(rng.random((N, 1) + shape, dtype=np.float32))
    lab = torch.tensor(rng.integers(0, ncls, (N,) + shape))
    y = torch.stack([(lab == c) for c in range(ncls)], 1).float()
    w = torch.tensor((rng.random((N, 1) + shape) > 0.25).astype(np.float32)).expand(N, ncls, *shape).contiguous() if weighted else None
    if w is not None:
        y = y * w
    axes = [0] + list(range(2, 2 + dim))
    loss_fn = getattr(metrics, loss_name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(lr=lr, num_classes=ncls, dim=dim, act_dtype='fp32', pretrained=False, loss_function=loss_fn)
    m.load_named(p)
    m = m.cuda()
    te = m.train_engine()
    assert isinstance(te, TrainEngineF32)
    out4, state = te.step_forward(X, y, w)
    flat, _ = te.step_backward(state)
    torch.cuda.synchronize()
    ws = state[0]
    dev = _device_acts(te, ws, N, dim)

    # ---- (0) the oracle on its own branch: loss, forward activations, mask ties
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p.items()}
    names = [k for k, t in pr.items() if t.requires_grad]
    acts = {}
    logits = _oracle_forward(pr, X, dim, acts=acts)
    assert torch.equal(logits, unet_ref.forward_logits(pr, X, dim=dim, training=True)), 'the restated forward IS the oracle forward'
    loss0 = loss_fn(torch.softmax(logits, 1), y, w, axes=axes)
    raw = dict(zip(names, torch.autograd.grad(loss0, [pr[k] for k in names])))
    assert abs(out4[0].item() - loss0.item()) <= 1e-6 * max(1.0, abs(loss0.item())), (out4[0].item(), loss0.item())
    fwd_err, flips = 0.0, 0
    for name, (pre, z, _, _) in acts.items():
        d = (dev[name] - z).abs().max().item()
        fwd_err = max(fwd_err, d)
        flip = (dev[name] > 0) != (z > 0)
        flips += int(flip.sum())
        if flip.any():
            assert pre[flip].abs().max().item() <= 4 * d + 1e-7, (name, 'a flipped ReLU bit away from a tie')
    assert fwd_err <= 1e-4, fwd_err

    # ---- (1) the oracle differentiated on the DEVICE's branch (its ReLU masks, its pool arg-max)
    pool = F.max_pool3d if dim == 3 else F.max_pool2d
    masks = {k: (t > 0).float() for k, t in dev.items()}
    pool_idx = [pool(dev[f'enc{l}.2'], 2, return_indices=True)[1] for l in range(te.levels - 1)]
    st = {}
    logits = _oracle_forward(pr, X, dim, masks=masks, pool_idx=pool_idx, acts=st)
    loss1 = loss_fn(torch.softmax(logits, 1), y, w, axes=axes)
    want = dict(zip(names, torch.autograd.grad(loss1, [pr[k] for k in names])))
    worst, worst_raw = ('', 0.0), ('', 0.0)
    for name in te.names:
        off, sz = te.offsets[name]
        got = flat[off:off + sz].cpu().reshape(want[name].shape)
        rel = (got - want[name]).abs().max().item() / max(want[name].abs().max().item(), 1e-12)
        rr = (got - raw[name]).abs().max().item() / max(raw[name].abs().max().item(), 1e-12)
        worst = max(worst, (name, rel), key=lambda t: t[1])
        worst_raw = max(worst_raw, (name, rr), key=lambda t: t[1])
        assert rel <= 1e-4, (name, rel)
    print(f'[fp32 train step {dim}-D {shape} {loss_name}] loss {out4[0].item():.6f} (oracle {loss0.item():.6f}); forward activations within '
          f'{fwd_err:.1e}, {flips} ReLU bits flipped at ties; gradients on the same branch: worst {worst[0]} off by {worst[1]:.1e} of its max; '
          f'raw (own branches): worst {worst_raw[0]} {worst_raw[1]:.1e}')
    assert worst_raw[1] <= 0.2

    # ---- (2) AdamW + running statistics
    new = {k: t.detach().clone() for k, t in pr.items()}
    mm = {k: torch.zeros_like(v) for k, v in new.items()}
    vv = {k: torch.zeros_like(t) for k, t in new.items()}
    # (AdamW's first step moves every entry by ~lr whatever the gradient's size, so it is applied to the DEVICE's gradients here: an
    #  entry with |g| ~ 1e-8 would turn the 1e-5 gradient noise into a different step; the kernel itself is what is under test)
    dev_g = {n: flat[te.offsets[n][0]:te.offsets[n][0] + te.offsets[n][1]].cpu().reshape(want[n].shape) for n in te.names}
    unet_ref.adamw_step(new, dev_g, mm, vv, 1, lr)
    cnt = X.numel() / X.shape[1]
    te.optimizer_step()
    torch.cuda.synchronize()
    for name, t in m.named_tensors().items():
        if unet_ref.is_buffer(name):
            prefix, bn, which = name.split('.')
            l = int(prefix[3:])
            _, _, mean, var = st[f'{prefix}.{bn[2]}']
            c = cnt / (2 ** (dim * l))
            ref = 0.9 * p[name] + 0.1 * (mean if which == 'running_mean' else var * c / (c - 1))
        else:
            ref = new[name]
        d = (t.detach().cpu() - ref).abs().max().item()
        assert d <= 2e-6 * max(1.0, ref.abs().max().item()), (name, d)
    # the Lightning-shaped API rides on the same engine
    loss2 = m.training_step((X, y, w))
    loss2.backward()
    assert m.tensor('head.weight').grad is not None
    ev = te.eval_step(X, y, w)
    assert np.isfinite(ev['Loss'])


@pytest.mark.parametrize('dim,shape,dtype', [(3, (32, 64, 64), 'bf16'), (2, (256, 256), 'fp16')])
def test_16_bit_step_against_the_fp32_form_on_the_device(dim, shape, dtype):
    """The fp32 form as the device-side checker of the 16-bit training path at a size where CPU autograd takes minutes: same weights,
    same batch, one step each.  With 10^5 .. 10^6 elements per channel the mask flips of the 16-bit storage noise average out, so
    the two gradients must point the same way tensor by tensor (the 32 x 48-pixel CPU comparison of test_gpu_train.py can only ask
    for cos > 0.85 on the deep tensors)."""
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    N, ncls = 2, 2
    p = unet_ref.init_params(dim=dim, ncls=ncls, seed=5)
    g = torch.Generator(device='cuda').manual_seed(0)
    X = torch.rand((N, 1) + tuple(s // 4 for s in shape), generator=g, device='cuda')
    X = F.interpolate(X, size=shape, mode='trilinear' if dim == 3 else 'bilinear', align_corners=False)
    lab = X[:, 0] > X.mean()
    y = torch.stack([~lab, lab], 1).float()
    w = (torch.rand((N, 1) + shape, generator=g, device='cuda') > 0.2).float().expand(N, ncls, *shape).contiguous()
    y = y * w
    grads = {}
    for name in ('fp32', dtype):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(lr=1e-3, num_classes=ncls, dim=dim, act_dtype=name, pretrained=False)
        m.load_named(p)
        m = m.cuda()
        te = m.train_engine()
        out4, state = te.step_forward(X, y, w)
        flat, _ = te.step_backward(state)
        torch.cuda.synchronize()
        grads[name] = (out4[0].item(), flat.clone(), dict(te.offsets), list(te.names))
        del te, m
    l32, g32, offs, names = grads['fp32']
    l16, g16, _, _ = grads[dtype]
    assert abs(l32 - l16) <= (2e-3 if dtype == 'fp16' else 1e-2), (l32, l16)
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    worst = min(((n, cos(g32[offs[n][0]:offs[n][0] + offs[n][1]], g16[offs[n][0]:offs[n][0] + offs[n][1]])) for n in names), key=lambda t: t[1])
    total = cos(g32, g16)
    print(f'[{dtype} vs fp32 form, {dim}-D {N} x {shape}] loss {l16:.5f} vs {l32:.5f}; cos(whole gradient) = {total:.5f}; worst tensor {worst[0]}: {worst[1]:.4f}')
    assert total >= (0.999 if dtype == 'fp16' else 0.99), total
    assert worst[1] >= (0.97 if dtype == 'fp16' else 0.8), worst          # the deepest level's small tensors (512 elements per channel) are the noisiest
