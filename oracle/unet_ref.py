"""Oracle: the canonical U-Net in plain torch CPU fp32 ops.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

**Parity unpinned at the smp boundary**: the reference builds its network with
``segmentation_models_pytorch==0.5.0`` (unet.py:33-61; pyproject.toml:19), which is
not vendored, not installed here, and has no reference test.  This file *defines*
the network the HIP path implements behind ``UNet(architecture='U-Net')``:

  levels L, channels base*2^l; stage = 2 x [Conv 3^d pad 1 no bias -> BatchNorm ->
  ReLU]; MaxPool 2 between encoder stages; decoder level l = ConvTranspose(k2,s2,
  bias) -> concat(skip, up) -> stage; head = Conv 1x1 (bias) -> softmax(dim=1)
  (softmax inside forward: unet.py:63-69).

Two evaluation modes:
* ``act_dtype=None``      exact fp32 everywhere (the definition);
* ``act_dtype=fp16/bf16`` same graph with the HIP path's rounding points restated:
  BatchNorm folded into the conv weights in fp32 then rounded to act_dtype, fp32
  accumulation, fp32 bias, ReLU, activations rounded to act_dtype after every
  stage conv / transposed conv; head in fp32 from act_dtype activations.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def channels(levels, base):
    return [base * (2 ** l) for l in range(levels)]


def param_shapes(dim=2, levels=4, base=32, cin=1, ncls=2):
    """Ordered {name: shape}.  The host module's state_dict uses the same names."""
    ch = channels(levels, base)
    k3, k2, k1 = (3,) * dim, (2,) * dim, (1,) * dim
    shapes = {}

    def stage(prefix, ci, co):
        for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
            shapes[f'{prefix}.conv{j}.weight'] = (b, a) + k3
            shapes[f'{prefix}.bn{j}.weight'] = (b,)
            shapes[f'{prefix}.bn{j}.bias'] = (b,)
            shapes[f'{prefix}.bn{j}.running_mean'] = (b,)
            shapes[f'{prefix}.bn{j}.running_var'] = (b,)
    for l in range(levels):
        stage(f'enc{l}', cin if l == 0 else ch[l - 1], ch[l])
    for l in range(levels - 2, -1, -1):
        shapes[f'dec{l}.up.weight'] = (ch[l + 1], ch[l]) + k2      # ConvTranspose: [in, out, k..]
        shapes[f'dec{l}.up.bias'] = (ch[l],)
        stage(f'dec{l}', 2 * ch[l], ch[l])                         # input = cat(skip, up)
    shapes['head.weight'] = (ncls, ch[0]) + k1
    shapes['head.bias'] = (ncls,)
    return shapes


def is_buffer(name):
    return name.endswith('running_mean') or name.endswith('running_var')


def init_params(dim=2, levels=4, base=32, cin=1, ncls=2, seed=0, randomize_bn=False):
    """He-normal conv weights; BN weight 1 / bias 0 / mean 0 / var 1 (or, with
    randomize_bn, non-trivial affine + running stats so that folding is exercised)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shp in param_shapes(dim, levels, base, cin, ncls).items():
        if name.endswith('conv1.weight') or name.endswith('conv2.weight') or name == 'head.weight':
            fan_in = shp[1] * math.prod(shp[2:])
            p[name] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith('up.weight'):
            fan_in = shp[0]                                       # each output sees C_in taps once
            p[name] = torch.randn(shp, generator=g) * math.sqrt(1.0 / fan_in)
        elif name.endswith('running_var'):
            p[name] = (0.5 + torch.rand(shp, generator=g)) if randomize_bn else torch.ones(shp)
        elif name.endswith('running_mean'):
            p[name] = (0.2 * torch.randn(shp, generator=g)) if randomize_bn else torch.zeros(shp)
        elif name.endswith('bn1.weight') or name.endswith('bn2.weight'):
            p[name] = (0.75 + 0.5 * torch.rand(shp, generator=g)) if randomize_bn else torch.ones(shp)
        elif name.endswith('bias'):
            p[name] = (0.1 * torch.randn(shp, generator=g)) if (randomize_bn or name == 'head.bias') \
                else torch.zeros(shp)
        else:
            raise KeyError(name)
    return p


def _conv(dim):
    return {2: F.conv2d, 3: F.conv3d}[dim]


def _convT(dim):
    return {2: F.conv_transpose2d, 3: F.conv_transpose3d}[dim]


def _pool(dim):
    return {2: F.max_pool2d, 3: F.max_pool3d}[dim]


def _rnd(t, act_dtype):
    return t if act_dtype is None else t.to(act_dtype).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """Storage rounding in BOTH directions: the value is rounded to act_dtype going forward
    and the gradient is rounded to act_dtype coming back -- the HIP training path keeps
    every activation and every activation gradient in act_dtype in HBM."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


def _rnd_ag(t, act_dtype):
    return t if act_dtype is None else _RoundBoth.apply(t, act_dtype)


def fold_bn(w, gamma, beta, mean, var):
    """Eval-mode BatchNorm folded into the preceding bias-free conv (fp32)."""
    a = gamma / torch.sqrt(var + BN_EPS)
    return w * a.view(-1, *([1] * (w.dim() - 1))), beta - mean * a


def fold_bn_exact(w, gamma, beta, mean, var):
    """fold_bn with every fp32 operation correctly rounded (numpy; torch's CPU sqrt / divide differ by an ulp):
    the quantised path below needs the fold bit-identical to the device's."""
    g, b, m, v = [t.detach().numpy().astype(np.float32) for t in (gamma, beta, mean, var)]
    a = g / np.sqrt(v + np.float32(BN_EPS))
    wf = w.detach().numpy().astype(np.float32) * a.reshape(-1, *([1] * (w.dim() - 1)))
    return torch.from_numpy(wf), torch.from_numpy(b - m * a)


def round_e4m3(x):
    """Nearest OCP e4m3 value (4 exponent bits, bias 7, 3 mantissa bits, subnormal step 2^-9), ties to even,
    for |x| <= 448.  numpy float32."""
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x)
    _, e = np.frexp(a)
    fl = np.where((a == 0) | (e - 1 < -6), -6, e - 1)
    step = np.ldexp(np.float32(1), fl - 3).astype(np.float32)
    return np.copysign(np.rint(a / step) * step, x).astype(np.float32)


def quantize_e4m3(w, out_axis=0):
    """Config C5's weight format: per-output-channel scale 2^k (k minimal with max|w| / 2^k <= 448) times an
    e4m3 value.  Returns the dequantised fp32 tensor (exactly representable in f16 and bf16)."""
    a = w.detach().numpy().astype(np.float32)
    red = tuple(i for i in range(a.ndim) if i != out_axis)
    amax = np.abs(a).max(axis=red, keepdims=True)
    m, e = np.frexp((amax / np.float32(448)).astype(np.float32))
    scale = np.where(amax > 0, np.ldexp(np.float32(1), np.where(m == 0.5, e - 1, e)), np.float32(1)).astype(np.float32)
    return torch.from_numpy((scale * round_e4m3(a / scale)).astype(np.float32))


def quantize_act_e4m3(t):
    """What the fp8 convolution does to its 16-bit input on the way into LDS: round to the nearest e4m3 value (ties to
    even), saturating at +-448."""
    a = np.clip(t.detach().numpy().astype(np.float32), -448.0, 448.0)
    return torch.from_numpy(round_e4m3(a))


def forward_logits(p, x, dim=2, levels=4, training=False, act_dtype=None, bn_stats_out=None, weight_quant=None, act_quant=False,
                   norm='batch', groups=8):
    """x: [N, cin, *spatial] float32 in [0,1].  Returns fp32 logits [N, ncls, *spatial].

    training=True uses batch statistics in BatchNorm (and, if bn_stats_out is a dict,
    records the (mean, biased var) it used per BN so running stats can be checked)."""
    conv, convT, pool = _conv(dim), _convT(dim), _pool(dim)
    x = _rnd(x, act_dtype)

    def stage(prefix, t):
        for j in (1, 2):
            w = p[f'{prefix}.conv{j}.weight']
            bn = [p[f'{prefix}.bn{j}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')]
            if norm == 'group':
                # GroupNorm(groups) variant (north_star "GroupNorm/BN"): statistics per (sample, group), the same at training
                # and inference -- nothing folds; rounding points of the HIP path: weights, raw conv output, stage output
                R = _rnd_ag if training else _rnd
                y = R(conv(t, R(w, act_dtype), padding=1), act_dtype)
                t = R(F.relu(F.group_norm(y, groups, bn[0], bn[1], eps=BN_EPS)), act_dtype)
            elif training:
                # act_dtype set: weights, raw conv output and the stage output are stored in
                # act_dtype (and so are their gradients), as in the HIP training path
                y = _rnd_ag(conv(t, _rnd_ag(w, act_dtype), padding=1), act_dtype)
                dims = [0] + list(range(2, y.dim()))
                mean = y.mean(dim=dims)
                var = y.var(dim=dims, unbiased=False)
                if bn_stats_out is not None:
                    bn_stats_out[f'{prefix}.bn{j}'] = (mean.detach(), var.detach())
                shape = [1, -1] + [1] * dim
                y = (y - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
                y = y * bn[0].view(shape) + bn[1].view(shape)
                t = _rnd_ag(F.relu(y), act_dtype)
            elif weight_quant == 'fp8_e4m3':
                wf, bf = fold_bn_exact(w, *bn)
                # act_quant: the stage convs (every conv but the first, Cin >= 32) run on the fp8 matrix cores -- their
                # 16-bit input is rounded to e4m3 as well (csrc/conv3_f8.hip); products of two e4m3 values are exact
                tin = quantize_act_e4m3(t) if (act_quant and w.shape[1] >= 32) else t
                t = _rnd(F.relu(conv(tin, quantize_e4m3(wf), bias=bf, padding=1)), act_dtype)
            else:
                wf, bf = fold_bn(w, *bn)
                t = _rnd(F.relu(conv(t, _rnd(wf, act_dtype), bias=bf, padding=1)), act_dtype)
        return t

    skips = []
    t = x
    for l in range(levels):
        t = stage(f'enc{l}', t)
        if l < levels - 1:
            skips.append(t)
            t = pool(t, 2)
    for l in range(levels - 2, -1, -1):
        R = _rnd_ag if training else _rnd
        wu = p[f'dec{l}.up.weight']
        if weight_quant == 'fp8_e4m3' and not training:
            wu = quantize_e4m3(wu, out_axis=1)
        up = R(convT(t, R(wu, act_dtype), bias=p[f'dec{l}.up.bias'], stride=2), act_dtype)
        t = stage(f'dec{l}', torch.cat([skips[l], up], dim=1))
    return conv(t, p['head.weight'], bias=p['head.bias'])


def forward(p, x, dim=2, levels=4, training=False, act_dtype=None, weight_quant=None, act_quant=False, norm='batch', groups=8):
    """Softmax probabilities NCHW(D), as UNet.forward returns them (unet.py:65-69)."""
    return torch.softmax(forward_logits(p, x, dim, levels, training, act_dtype, weight_quant=weight_quant,
                                        act_quant=act_quant, norm=norm, groups=groups), dim=1)


def flops_per_voxel(dim=2, levels=4, base=32, cin=1, ncls=2):
    """Algorithmic forward FLOPs (2*MAC) per full-resolution voxel; SURVEY.md 8(d)."""
    ch = channels(levels, base)
    taps = 3 ** dim
    f = 0.0
    for l in range(levels):
        scale = 1.0 / (2 ** (dim * l))
        ci = cin if l == 0 else ch[l - 1]
        f += scale * 2 * taps * (ci * ch[l] + ch[l] * ch[l])
    for l in range(levels - 2, -1, -1):
        scale = 1.0 / (2 ** (dim * l))
        f += scale * 2 * ch[l + 1] * ch[l]                         # convT: one tap per output voxel
        f += scale * 2 * taps * (2 * ch[l] * ch[l] + ch[l] * ch[l])
    f += 2 * ch[0] * ncls
    return f


def adamw_step(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults restated (unet.py:71-73); in-place on fp32 tensors."""
    for k in params:
        if is_buffer(k):
            continue
        g = grads[k]
        params[k].mul_(1 - lr * wd)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)
