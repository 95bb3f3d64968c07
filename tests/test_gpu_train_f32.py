"""fp32 parity form of the training step (interactive_unet/train_engine_f32.py, csrc/train_f32.hip + precise_f32.hip) against CPU
autograd: the oracle's forward in training mode (oracle/unet_ref.py, plain fp32) + the host metrics' loss + torch.autograd +
the restated AdamW.  The device step differs only by the order of its fp32 sums, so the bar is tight -- it replaces the cosine gate
that the 16-bit training path can be held to (tests/test_gpu_train.py):

  loss                         |native - oracle| <= 1e-6 (relative to max(1, |loss|))
  every parameter gradient     max |native - oracle| <= 1e-4 x max |oracle| per tensor  ("<= 1e-4 relative")
  parameters after AdamW       max |native - oracle| <= 1e-6
  BatchNorm running statistics max |native - oracle| <= 1e-6

and the kernel-level pieces (weight gradient, BatchNorm backward, max-pool backward) are checked on their own first."""
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


def _oracle_step(p, X, y, w, dim, loss_name, lr):
    from interactive_unet import metrics as host_metrics
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p.items()}
    st = {}
    logits = unet_ref.forward_logits(pr, X, dim=dim, training=True, bn_stats_out=st)
    probs = torch.softmax(logits, 1)
    loss = getattr(host_metrics, loss_name)(probs, y, w, axes=[0] + list(range(2, 2 + dim)))
    names = [k for k, t in pr.items() if t.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [pr[k] for k in names])))
    new = {k: t.detach().clone() for k, t in pr.items()}
    m = {k: torch.zeros_like(v) for k, v in new.items()}
    v = {k: torch.zeros_like(t) for k, t in new.items()}
    unet_ref.adamw_step(new, grads, m, v, 1, lr)
    cnt = X.numel() / X.shape[1]
    for name, (mean, var) in st.items():
        new[name + '.running_mean'] = 0.9 * p[name + '.running_mean'] + 0.1 * mean
        new[name + '.running_var'] = 0.9 * p[name + '.running_var'] + 0.1 * var * cnt / (cnt - 1)
    return loss.item(), grads, new


@pytest.mark.parametrize('dim,shape,ncls,loss_name,weighted', [
    (2, (32, 48), 2, 'mcc_ce_loss', True),
    (2, (16, 16), 3, 'dice_ce_loss', False),
    (3, (8, 16, 16), 2, 'mcc_ce_loss', True),
    (3, (8, 8, 24), 4, 'iou_loss', True),
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
    want_loss, want_g, want_p = _oracle_step(p, X, y, w, dim, loss_name, lr)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(lr=lr, num_classes=ncls, dim=dim, act_dtype='fp32', pretrained=False, loss_function=getattr(metrics, loss_name))
    m.load_named(p)
    m = m.cuda()
    te = m.train_engine()
    assert isinstance(te, TrainEngineF32)
    out4, state = te.step_forward(X, y, w)
    loss = out4[0].item()
    assert abs(loss - want_loss) <= 1e-6 * max(1.0, abs(want_loss)), (loss, want_loss)
    flat, _ = te.step_backward(state)
    worst = ('', 0.0)
    for name in te.names:
        off, sz = te.offsets[name]
        got = flat[off:off + sz].cpu().reshape(want_g[name].shape)
        rel = (got - want_g[name]).abs().max().item() / max(want_g[name].abs().max().item(), 1e-12)
        if rel > worst[1]:
            worst = (name, rel)
        assert rel <= 1e-4, (name, rel)
    print(f'[fp32 train step {dim}-D {shape} {loss_name}] loss {loss:.6f} (oracle {want_loss:.6f}); worst gradient: {worst[0]} off by {worst[1]:.2e} of its max')
    te.optimizer_step()
    torch.cuda.synchronize()
    for name, t in m.named_tensors().items():
        d = (t.detach().cpu() - want_p[name]).abs().max().item()
        assert d <= 1e-6 * max(1.0, want_p[name].abs().max().item()), (name, d)
    # the Lightning-shaped API rides on the same engine
    loss2 = m.training_step((X, y, w))
    loss2.backward()
    assert m.tensor('head.weight').grad is not None
    ev = te.eval_step(X, y, w)
    assert np.isfinite(ev['Loss'])
