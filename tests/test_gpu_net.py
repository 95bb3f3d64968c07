"""End-to-end parity of the native U-Net forward and the device predict kernels against
the oracle (oracle/unet_ref.py, oracle/predict_ref.py) and the reference goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref, predict_ref


def _engine(dim, ncls=2, dtype=torch.float16, seed=0, cin=1):
    from interactive_unet.engine import Engine
    p = unet_ref.init_params(dim=dim, cin=cin, ncls=ncls, seed=seed, randomize_bn=True)
    e = Engine(dim=dim, cin=cin, ncls=ncls, act_dtype=dtype)
    e.load_eval({k: v.cuda() for k, v in p.items()})
    return e, p


def _smooth(shape, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), 3)
    v = (v - v.min()) / (v.max() - v.min())
    return (v * 255).astype(np.uint8)


def _stub(kind):
    def softmax(l):
        e = np.exp(l - l.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)

    def f(x):
        x = x.astype(np.float32)
        B, _, H, W = x.shape
        if kind == 0:
            return softmax(np.concatenate([x, 1 - x], 1))
        r = (np.arange(H, dtype=np.float32) / H).reshape(1, 1, H, 1)
        c = (np.arange(W, dtype=np.float32) / W).reshape(1, 1, 1, W)
        return softmax(np.concatenate([x * (1 + r), x * (0.5 + 2 * c) - 0.3 * r, 0.2 + 0 * x], 1))
    return f


@pytest.mark.parametrize('dim,shape,dtype', [(2, (64, 96), torch.float16), (2, (128, 128), torch.bfloat16),
                                             (3, (16, 32, 48), torch.float16), (3, (32, 32, 32), torch.bfloat16)])
def test_forward_logits_vs_oracle(dim, shape, dtype):
    """16-bit modes at small shapes against TWO oracles (the north-star gate -- 1e-3 absolute against the fp32 CPU path --
    is tests/test_gpu_parity.py, on the fp32 parity mode):
    (a) the oracle evaluated with the SAME rounding points (act_dtype storage, fp32 accumulate): max |dlogit| <=
        3e-3 x scale (fp16) / 2.4e-2 x scale (bf16), rms <= 6e-4 x scale (x 8 for bf16), scale = max |logit| -- this pins
        the kernels (fragment maps, epilogues, folds), not the storage precision;
    (b) the pure-fp32 oracle: asserted bound max |dlogit| <= 4e-3 x scale (fp16) / 3e-2 x scale (bf16), the measured
        deviation of 16-bit activation storage through 18 convs with 2x headroom (assert it, do not just print it).
    The class map is integer-exact wherever the same-rounding oracle's top-2 margin exceeds twice tolerance (a)."""
    e, p = _engine(dim, ncls=3, dtype=dtype, seed=1)
    N = 2
    img = np.stack([_smooth(shape, 10 + i) for i in range(N)])[:, None]        # N,1,*shape uint8
    x = torch.tensor(img)
    xf = x.float() / 255.0
    ref = unet_ref.forward_logits(p, xf, dim=dim, act_dtype=dtype)
    ref32 = unet_ref.forward_logits(p, xf, dim=dim)
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    logits = torch.empty((N, 3) + shape, device='cuda')
    probs = torch.empty((N, 3) + shape, device='cuda')
    cls = torch.empty(N, vox, dtype=torch.uint8, device='cuda')
    xd = x.cuda()
    e.infer(xd, (vox, vox, H * W, W, 1), N, D, H, W, logits=logits, probs=probs, cls=cls)
    torch.cuda.synchronize()
    got = logits.cpu()
    err = (got - ref).abs().max().item()
    err32 = (got - ref32).abs().max().item()
    print(f'dim={dim} {dtype}: max|logit - oracle(same rounding)| = {err:.2e}, vs fp32 oracle = {err32:.2e}, '
          f'logit scale = {ref32.abs().max().item():.2f}')
    # Tolerance.  fp16/bf16 activation storage makes the logits of an 18-conv network chaotic
    # at the level of the storage ulp: the ORACLE ITSELF moves by 1.3e-3 x scale (max) /
    # 2.4e-4 x scale (rms) in fp16 when only its accumulation is switched fp32 -> fp64 at
    # identical rounding points (measured, DESIGN.md "Parity"); bf16 is 8x coarser.  So the
    # 1e-3 of north_star is applied relative to the logit scale, with 3x headroom on the max.
    scale = max(1.0, ref32.abs().max().item())
    ulp = 1.0 if dtype == torch.float16 else 8.0
    tol = 3e-3 * ulp * scale
    rms = (got - ref).pow(2).mean().sqrt().item()
    print(f'   rms = {rms:.2e}, tol(max) = {tol:.2e}')
    assert err <= tol
    assert rms <= 6e-4 * ulp * scale
    assert err32 <= (4e-3 if dtype == torch.float16 else 3e-2) * scale, (err32, scale)      # (b): storage-precision bound vs fp32
    top2 = torch.topk(ref, 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).reshape(N, vox)
    sure = margin > 2 * tol
    want = ref.argmax(1).reshape(N, vox)
    assert sure.float().mean() > 0.8
    assert torch.equal(cls.cpu().long()[sure], want[sure])
    assert (probs.cpu() - torch.softmax(ref, 1)).abs().max() <= tol


def test_infer_strided_2p5d_matches_oracle_predict_block():
    """predict.py:79-112 on device: the three slice orientations are strided views of one
    uint8 block; the head accumulates into [S,S,S,C] with permuted strides and divides by 3."""
    e, p = _engine(2, ncls=2, dtype=torch.float16, seed=2)
    S = 32
    blk = _smooth((S, S, S), 5)
    out = torch.zeros(S, S, S, 2, device='cuda')
    bd = torch.tensor(blk).cuda()
    C = 2
    sz, sy, sx = S * S, S, 1
    oz, oy, ox = S * S * C, S * C, C
    views = [((sz, 0, 0, sy, sx), (oz, 1, 0, oy, ox)),      # slices along z: rows y, cols x
             ((sy, 0, 0, sz, sx), (oy, 1, 0, oz, ox)),      # along y: rows z, cols x
             ((sx, 0, 0, sz, sy), (ox, 1, 0, oz, oy))]      # along x: rows z, cols y
    for i, (xs, os_) in enumerate(views):
        e.infer(bd, xs, S, 1, S, S, probs=out, out_strides=os_, accumulate=(i > 0), divisor=3.0 if i == 2 else 1.0)
    torch.cuda.synchronize()

    def model_fn(batch):
        return unet_ref.forward(p, torch.tensor(batch), dim=2, act_dtype=torch.float16).numpy()
    want = predict_ref.predict_block(model_fn, blk.astype(np.float32) / 255.0, num_classes=2, batch_size=8)
    assert np.abs(out.cpu().numpy() - want).max() <= 3e-3      # probabilities; see tolerance note above


def test_gather_blend_quantise_bit_exact(golden_dir):
    """get_padded_block / blend / normalise+quantise on device vs the reference goldens
    (integer results bit for bit)."""
    from interactive_unet import _native as nv
    g = np.load(os.path.join(golden_dir, 'predict.npz'))
    vol = g['pad_vol']
    vd = torch.tensor(vol).cuda()
    for i, c in enumerate(g['pad_coords']):
        S = int(c[3] - c[0])
        out = torch.empty(S, S, S, dtype=torch.uint8, device='cuda')
        nv.call('iunet_gather_block', nv.ptr(vd), *vol.shape, int(c[0]), int(c[1]), int(c[2]), S, nv.ptr(out), nv.stream())
        assert np.array_equal(out.cpu().numpy(), g[f'pad{i}'])
    # blend with the same stub probabilities the golden used (computed on the host here)
    volume = g['blend_volume']
    V = volume.shape
    S, C = 32, 3
    window = predict_ref.gaussian_3d(S)
    wd = torch.tensor(window).cuda()
    pred = torch.zeros(V + (C,), device='cuda')
    weight = torch.zeros(V, device='cuda')
    vold = torch.tensor(volume).cuda()
    bc, pbc, lbc = predict_ref.get_block_coordinates(V, S, 0.25)
    for b, pb, lb in zip(bc, pbc, lbc):
        blk = torch.empty(S, S, S, dtype=torch.uint8, device='cuda')
        nv.call('iunet_gather_block', nv.ptr(vold), *V, int(pb[0]), int(pb[1]), int(pb[2]), S, nv.ptr(blk), nv.stream())
        P = predict_ref.predict_block(_stub(1), blk.cpu().numpy().astype('float32') / 255.0, C, 8, (0, 1, 2))
        Pd = torch.tensor(P).cuda()
        nv.call('iunet_blend_accumulate', nv.ptr(pred), nv.ptr(weight), nv.ptr(Pd), nv.ptr(wd), *V, C, S,
                nv.int_array(b), nv.int_array(lb), nv.stream())
    final = torch.empty(V + (C,), dtype=torch.uint8, device='cuda')
    nv.call('iunet_normalize_quantize', nv.ptr(pred), nv.ptr(weight), nv.ptr(final), int(np.prod(V)), C, 1e-3, nv.stream())
    torch.cuda.synchronize()
    assert np.array_equal(weight.cpu().numpy(), g['blend_weight'])
    of, _, _ = predict_ref.blend_volume(volume, lambda blk: predict_ref.predict_block(_stub(1), blk, C, 8, (0, 1, 2)), S, C)
    assert np.array_equal(final.cpu().numpy(), of)          # same inputs -> bit-exact vs the oracle
    d = np.abs(final.cpu().numpy().astype(int) - g['blend_final'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3            # golden used torch's softmax for the stub


def test_c5_shape_5_level_base64_4_class_forward_and_step():
    """BASELINE.json configs[4] architecture (3-D, 5 levels, base 64, 4 classes) in bf16 activations:
    forward parity with the oracle and one native training step (architecture generality of the engine:
    levels, base, head width; the fp8-weight variant is test_c5_fp8_weights_forward below)."""
    import warnings
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    dim, levels, base, ncls = 3, 5, 64, 4
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, ncls=ncls, seed=4, randomize_bn=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=ncls, dim=dim, levels=levels, base=base, act_dtype='bf16', pretrained=False)
    m.load_named(p)
    m = m.cuda().eval()
    shape = (32, 32, 48)
    x = torch.tensor(_smooth(shape, 3))[None, None]
    probs = m(x.cuda()).cpu()
    ref = unet_ref.forward(p, x.float() / 255.0, dim=dim, levels=levels, act_dtype=torch.bfloat16)
    err = (probs - ref).abs().max().item()
    print(f'C5-shape forward: max |prob - oracle| = {err:.2e}')
    assert err < 3e-2                                   # bf16 storage through 22 convs (see tolerance note above)
    lab = torch.tensor(_smooth(shape, 4))[None] // 64
    y = torch.stack([(lab == c) for c in range(ncls)], 1).float()
    te = TrainEngine(m, lr=1e-3, loss_kind='dice_ce')
    out1 = te.train_step(x, y, None)
    out2 = te.train_step(x, y, None)
    assert np.isfinite(out1['Loss']) and np.isfinite(out2['Loss']) and out2['Loss'] < out1['Loss'] + 0.05


def test_2p5d_block_as_one_batch_of_views_equals_the_three_forwards(monkeypatch):
    """predict.predict_block_device in the default mode: the three axes' slices as ONE batch of 3 S (EngineX2.infer_views: per-view first
    conv and per-view fused head, the network once) against the three forwards of S slices -- the same bits, and both within tolerance
    of the oracle's predict_block (predict.py:79-112)."""
    import warnings
    from interactive_unet import predict as P
    from interactive_unet.unet import UNet
    S, C = 64, 2
    p = unet_ref.init_params(dim=2, ncls=C, seed=3, randomize_bn=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=C, dim=2, pretrained=False)
    m.load_named(p)
    m = m.cuda().eval()
    blk = torch.tensor(_smooth((S, S, S), 8)).cuda()
    outs = {}
    for mode in ('views', 'sequential'):
        if mode == 'sequential':
            monkeypatch.setenv('IUNET_2P5D_SEQUENTIAL', '1')
        out = torch.full((S, S, S, C), float('nan'), device='cuda')
        for _ in range(2):                      # (second call: the engines' C++ graph where it applies)
            P.predict_block_device(m, blk, out, C, None, (0, 1, 2))
        outs[mode] = out.clone()
    assert m.engine('eval').form == 'x2m'
    assert torch.equal(outs['views'], outs['sequential'])
    # two axes, and a batch size below S (falls back to the loop): still the oracle's numbers
    out2 = torch.empty((S, S, S, C), device='cuda')
    monkeypatch.delenv('IUNET_2P5D_SEQUENTIAL')
    P.predict_block_device(m, blk, out2, C, None, (0, 2))
    fn = lambda b: unet_ref.forward(p, torch.tensor(b), dim=2).numpy()
    x = blk.cpu().numpy().astype(np.float32) / 255.0
    want3 = predict_ref.predict_block(fn, x, C, 8, (0, 1, 2))
    want2 = predict_ref.predict_block(fn, x, C, 8, (0, 2))
    assert np.abs(outs['views'].cpu().numpy() - want3).max() <= 2e-4
    assert np.abs(out2.cpu().numpy() - want2).max() <= 2e-4
    out3 = torch.empty((S, S, S, C), device='cuda')
    P.predict_block_device(m, blk, out3, C, 16, (0, 1, 2))
    assert torch.equal(out3, outs['views'])
