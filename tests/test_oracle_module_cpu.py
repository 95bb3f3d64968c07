"""The oracle's network against an INDEPENDENT torch.nn module ("a plain
torch.nn CPU module").

oracle/unet_ref.forward_logits evaluates eval-mode BatchNorm in the FOLDED form (fold_bn, then a conv with bias) -- the operation
order the device uses.  Here the same named parameters are loaded into a module built from nn.Conv{2,3}d, nn.BatchNorm{2,3}d (its own
eval / train arithmetic), nn.GroupNorm, nn.ReLU, nn.MaxPool{2,3}d and nn.ConvTranspose{2,3}d, composed as SURVEY.md section 8d states
the canonical network, and the two must agree to fp32 rounding: the oracle is then more than its own folded form.
"""
import pytest
import torch
import torch.nn as nn

from oracle import unet_ref


class _Stage(nn.Module):
    def __init__(self, dim, ci, co, norm, groups):
        super().__init__()
        Conv = {2: nn.Conv2d, 3: nn.Conv3d}[dim]
        BN = {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dim]
        mk = (lambda c: BN(c, eps=unet_ref.BN_EPS, momentum=unet_ref.BN_MOMENTUM)) if norm == 'batch' else (lambda c: nn.GroupNorm(groups, c, eps=unet_ref.BN_EPS))
        self.conv1, self.bn1 = Conv(ci, co, 3, padding=1, bias=False), mk(co)
        self.conv2, self.bn2 = Conv(co, co, 3, padding=1, bias=False), mk(co)

    def forward(self, t):
        t = torch.relu(self.bn1(self.conv1(t)))
        return torch.relu(self.bn2(self.conv2(t)))


class _Dec(_Stage):
    def __init__(self, dim, c_below, c, norm, groups):
        super().__init__(dim, 2 * c, c, norm, groups)
        self.up = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dim](c_below, c, 2, stride=2, bias=True)


class PlainUNet(nn.Module):
    """SURVEY.md 8d, written with nn modules only; parameter names = oracle/unet_ref.param_shapes (encL.convJ / bnJ, decL.up, head)."""

    def __init__(self, dim=2, levels=4, base=32, cin=1, ncls=2, norm='batch', groups=8):
        super().__init__()
        ch = [base * 2 ** l for l in range(levels)]
        self.levels = levels
        self.pool = {2: nn.MaxPool2d, 3: nn.MaxPool3d}[dim](2)
        for l in range(levels):
            setattr(self, f'enc{l}', _Stage(dim, cin if l == 0 else ch[l - 1], ch[l], norm, groups))
        for l in range(levels - 1):
            setattr(self, f'dec{l}', _Dec(dim, ch[l + 1], ch[l], norm, groups))
        self.head = {2: nn.Conv2d, 3: nn.Conv3d}[dim](ch[0], ncls, 1, bias=True)

    def forward(self, x):
        skips, t = [], x
        for l in range(self.levels):
            t = getattr(self, f'enc{l}')(t)
            if l < self.levels - 1:
                skips.append(t)
                t = self.pool(t)
        for l in range(self.levels - 2, -1, -1):
            d = getattr(self, f'dec{l}')
            t = _Stage.forward(d, torch.cat([skips[l], d.up(t)], dim=1))
        return self.head(t)


def _load(m, p, norm):
    sd = {k: v.clone() for k, v in p.items() if norm == 'batch' or not unet_ref.is_buffer(k)}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith('num_batches_tracked') for k in missing), missing      # every tensor of the oracle has a home, and vice versa


CASES = [(2, 4, 32, 1, 2, (64, 96)), (2, 5, 32, 3, 4, (64, 64)), (3, 4, 32, 1, 2, (16, 32, 24)), (3, 5, 32, 2, 3, (32, 16, 16))]


@pytest.mark.parametrize('dim,levels,base,cin,ncls,shape', CASES)
@pytest.mark.parametrize('norm', ['batch', 'group'])
def test_eval_forward_equals_the_plain_module(dim, levels, base, cin, ncls, shape, norm):
    torch.manual_seed(0)
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, cin=cin, ncls=ncls, seed=11, randomize_bn=True)
    m = PlainUNet(dim, levels, base, cin, ncls, norm).eval()
    _load(m, p, norm)
    x = torch.rand((2, cin) + shape)
    with torch.no_grad():
        want = m(x)
        got = unet_ref.forward_logits(p, x, dim=dim, levels=levels, norm=norm)
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print(f'[oracle vs nn.Module] {dim}-D {levels} levels {norm}: max |diff| {err:.2e} at logit scale {scale:.2f}')
    assert err <= 1e-5 * max(1.0, scale)


@pytest.mark.parametrize('dim,levels,base,cin,ncls,shape', CASES[::2])
def test_training_forward_and_gradients_equal_the_plain_module(dim, levels, base, cin, ncls, shape):
    """Batch statistics (nn.BatchNorm in train mode) and autograd through both: logits, every parameter gradient, and the running
    statistics the module updates against the (mean, biased variance) the oracle reports.  In float64: two correct fp32 evaluations
    differ by ~1e-5 in a pre-activation, which flips a ReLU / max-pool mask bit wherever a value sits inside that noise and moves a
    gradient SUM by 1e-3 of its size (training parity); in double the same composition agrees to 1e-9."""
    p = {k: v.double() for k, v in unet_ref.init_params(dim=dim, levels=levels, base=base, cin=cin, ncls=ncls, seed=5, randomize_bn=True).items()}
    m = PlainUNet(dim, levels, base, cin, ncls, 'batch').double().train()
    _load(m, p, 'batch')
    torch.manual_seed(1)
    x = torch.rand((2, cin) + shape, dtype=torch.float64)
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p.items()}
    stats = {}
    got = unet_ref.forward_logits(pr, x, dim=dim, levels=levels, training=True, bn_stats_out=stats)
    want = m(x)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-9 * max(1.0, scale)
    tgt = torch.rand_like(want)
    (got * tgt).sum().backward()
    (want * tgt).sum().backward()
    for k, t in m.named_parameters():
        g, w = pr[k].grad, t.grad
        assert (g - w).abs().max().item() <= 1e-8 * max(1e-6, w.abs().max().item()), k
    n = x.numel() // cin
    for k, (mean, var) in stats.items():
        bn = m.get_submodule(k)
        mom = unet_ref.BN_MOMENTUM
        want_mean = (1 - mom) * p[f'{k}.running_mean'] + mom * mean
        cnt = n // (2 ** (dim * int(k[3])))
        want_var = (1 - mom) * p[f'{k}.running_var'] + mom * var * cnt / (cnt - 1)
        assert torch.allclose(bn.running_mean, want_mean, rtol=1e-9, atol=1e-12), k
        assert torch.allclose(bn.running_var, want_var, rtol=1e-9, atol=1e-12), k
