"""C4 (BASELINE.json configs[3]) at full size on one GPU: a 1024^3 uint8 volume, 128^3 blocks, overlap 0.25 ->
11^3 = 1 331 blocks through the native 3-D network.  A CPU oracle of the whole volume is out of reach (1.2 PFLOP),
so the run is checked through size-independent properties of predict.py:201-256 plus an exact oracle comparison
of the one region that a single block covers."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref, predict_ref


def _box_sums(win, boxes):
    """sum of win[z0:z1, y0:y1, x0:x1] for every box, through a float64 summed-area table."""
    sat = np.zeros(tuple(s + 1 for s in win.shape), dtype=np.float64)
    sat[1:, 1:, 1:] = win.astype(np.float64).cumsum(0).cumsum(1).cumsum(2)
    tot = 0.0
    for z0, y0, x0, z1, y1, x1 in boxes:
        tot += (sat[z1, y1, x1] - sat[z0, y1, x1] - sat[z1, y0, x1] - sat[z1, y1, x0]
                + sat[z0, y0, x1] + sat[z0, y1, x0] + sat[z1, y0, x0] - sat[z0, y0, x0])
    return tot


def test_c4_full_size_volume_properties():
    from interactive_unet import predict
    from interactive_unet.unet import UNet
    S, C, V = 128, 2, (1024, 1024, 1024)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(num_classes=C, dim=3, act_dtype='fp16', pretrained=False)
    p = unet_ref.init_params(dim=3, ncls=C, seed=5, randomize_bn=True)
    model.load_named(p)
    model = model.cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(2)
    vol = torch.randint(0, 256, V, dtype=torch.uint8, device='cuda', generator=g)       # rng(2), generated on device
    bc, pbc, lbc = predict.get_block_coordinates(np.array(V), input_size=S, overlap=0.25)
    assert len(pbc) == 1331 and tuple(pbc[0][:3]) == (-32, -32, -32) and tuple(pbc[-1][3:]) == (1056, 1056, 1056)

    import time
    torch.cuda.synchronize()
    t0 = time.time()
    acc = predict.predict_volume_array(model, vol, input_size=S, num_classes=C, overlap=0.25, finalize=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'C4 on one GPU: 1331 blocks in {dt:.2f} s = {1024 ** 3 / dt / 1e6:.0f} M volume voxels/s '
          f'({1331 * 128 ** 3 / dt / 1e6:.0f} M processed voxels/s), first call (includes workspace allocation)')

    # (1) checksum of the blend weights: sum over the volume == sum over blocks of the window over its local box
    win = predict_ref.gaussian_3d(S)
    want = _box_sums(win, [tuple(int(v) for v in l) for l in lbc])
    got = acc.weight.sum(dtype=torch.float64).item()
    assert abs(got - want) <= 1e-6 * want, (got, want)
    assert acc.weight.min().item() >= 1e-3                     # every voxel is covered (window floor, predict.py:343)

    # (2) softmax partition of unity survives the blend: sum_c pred == weight (fp32 accumulation error only)
    zs = slice(448, 576)                                       # a 128-plane slab is enough (and fits comfortably)
    dev = (acc.pred[zs].sum(-1) - acc.weight[zs]).abs().max().item()
    assert dev <= 1e-5 * acc.weight[zs].max().item() * 27, dev

    out = acc.finalize()
    assert out.shape == V + (C,) and out.dtype == torch.uint8
    s = out[zs].sum(-1, dtype=torch.int32)
    assert s.min().item() >= 253 and s.max().item() <= 255     # truncating cast: each class loses < 1 LSB

    # (3) the corner [0, 64)^3 is covered by block 0 alone (block 1 starts at 64): result == uint8(255 * P_block0)
    corner = vol[:96, :96, :96].cpu().numpy()
    blk = predict_ref.get_padded_block(corner, *[int(v) for v in pbc[0]])
    assert blk.shape == (S, S, S)
    torch.manual_seed(0)
    pr = unet_ref.forward(p, torch.tensor(blk.astype(np.float32) / 255)[None, None], dim=3, act_dtype=torch.float16)
    want_c = (255 * pr[0].permute(1, 2, 3, 0).numpy()[32:96, 32:96, 32:96]).astype(np.uint8)
    got_c = out[:64, :64, :64].cpu().numpy()
    d = np.abs(got_c.astype(int) - want_c.astype(int))
    print(f'C4 corner vs oracle: max |uint8 diff| = {d.max()}, differing = {(d > 0).mean():.4f}')
    assert d.max() <= 2


def test_c4_full_size_pyramid():
    """The multiscale pyramid of a 1024^3 uint8 volume (utils.py:50-77: 128^3 chunks, 256^3 shards -> levels 512^3, 256^3,
    128^3) on the device.  Checked exactly against the scipy-pinned oracle on sampled shards of every level, plus the
    size-independent property that with 256^3 blocks of an even size nothing is constant-filled: every level is a pure
    gather, so its value histogram support is a subset of the source's and level l only holds voxels of level l - 1."""
    from interactive_unet import utils
    from oracle import multiscale_ref as mr
    V = (1024, 1024, 1024)
    g = torch.Generator(device='cuda').manual_seed(7)
    vol = torch.randint(1, 256, V, dtype=torch.uint8, device='cuda', generator=g)        # no zeros: a constant fill would show
    levels = utils.multiscale_levels(vol, (128,) * 3, (256,) * 3)
    assert [tuple(l.shape) for l in levels] == [(512,) * 3, (256,) * 3, (128,) * 3]
    src = vol
    rng = np.random.default_rng(3)
    for lv in levels:
        assert int((lv == 0).sum()) == 0
        nb = src.shape[0] // 256
        picks = {(0, 0, 0), (nb - 1, nb - 1, nb - 1)} | {tuple(rng.integers(0, nb, 3)) for _ in range(2)}
        for bi, bj, bk in picks:
            blk = src[bi * 256:(bi + 1) * 256, bj * 256:(bj + 1) * 256, bk * 256:(bk + 1) * 256].cpu().numpy()
            want = mr.zoom_nearest(blk, 0.5)
            got = lv[bi * 128:(bi + 1) * 128, bj * 128:(bj + 1) * 128, bk * 128:(bk + 1) * 128].cpu().numpy()
            assert np.array_equal(got, want), (tuple(lv.shape), bi, bj, bk)
        src = lv


def test_c4_subvolume_throughput_mode_vs_parity_mode():
    """"parity sub-check on a 256^3 corner".  A CPU oracle of 27 blocks of 128^3 takes minutes; the fp32 parity
    mode (held within 1e-5 of the CPU oracle at 128^3 by tests/test_gpu_parity.py) stands in for it on the device: the same
    256^3 volume through the whole prediction pipeline (block grid, reflect blocks, forward, Gaussian blend, truncating
    quantisation) in fp16 and in fp32.  The uint8 probabilities may differ by the 16-bit storage noise (<= 3e-3 -> < 1 LSB)
    plus the truncation: at most 2 LSB anywhere; the class map is equal wherever the classes are more than 4 LSB apart."""
    from interactive_unet import predict
    from interactive_unet.unet import UNet
    S, C, V = 128, 2, (256, 256, 256)
    p = unet_ref.init_params(dim=3, ncls=C, seed=5, randomize_bn=True)
    outs = {}
    g = torch.Generator(device='cuda').manual_seed(2)
    vol = torch.randint(0, 256, V, dtype=torch.uint8, device='cuda', generator=g)
    for dt in ('fp16', 'fp32'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = UNet(num_classes=C, dim=3, act_dtype=dt, pretrained=False)
        model.load_named(p)
        model = model.cuda().eval()
        outs[dt] = predict.predict_volume_array(model, vol, input_size=S, num_classes=C, overlap=0.25).cpu().numpy().astype(int)
        del model
        torch.cuda.empty_cache()
    d = np.abs(outs['fp16'] - outs['fp32'])
    sure = np.abs(outs['fp32'][..., 0] - outs['fp32'][..., 1]) > 4
    print(f'C4 sub-volume 256^3 (27 blocks): fp16 vs fp32 parity mode max |uint8 diff| = {d.max()}, differing = {(d > 0).mean():.4f}, '
          f'class map compared on {sure.mean():.3f} of the voxels')
    assert d.max() <= 2
    assert np.array_equal(outs['fp16'].argmax(-1)[sure], outs['fp32'].argmax(-1)[sure])


def test_c4_subvolume_default_mode_vs_parity_mode():
    """The same 256^3 pipeline in the DEFAULT prediction mode -- what UNet() selects (engine_auto.EngineAuto: x2m here) -- against the
    fp32 parity mode: the probabilities differ by <= 3e-4, so the truncated uint8 outputs differ by at most 1 LSB
    (a value on either side of an integer), and the class map of the uint8 result is equal wherever the fp32 mode's two classes are more
    than 2 LSB apart (each class may move by one: (126, 128) against (127, 127) is a tie made by the truncation, not a class change);
    plus the accumulators before quantisation: |pred / weight| within 1e-3 everywhere (blend of probabilities each within tolerance)."""
    from interactive_unet import predict
    from interactive_unet.unet import UNet
    S, C, V = 128, 2, (256, 256, 256)
    p = unet_ref.init_params(dim=3, ncls=C, seed=5, randomize_bn=True)
    g = torch.Generator(device='cuda').manual_seed(2)
    vol = torch.randint(0, 256, V, dtype=torch.uint8, device='cuda', generator=g)
    outs, probs, form = {}, {}, None
    for dt in (None, 'fp32'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = UNet(num_classes=C, dim=3, act_dtype=dt, pretrained=False)
        model.load_named(p)
        model = model.cuda().eval()
        acc = predict.predict_volume_array(model, vol, input_size=S, num_classes=C, overlap=0.25, finalize=False)
        probs[dt] = (acc.pred / acc.weight.clamp_min(1e-3)[..., None]).cpu()
        outs[dt] = acc.finalize().cpu().numpy().astype(int)
        if dt is None:
            eng = model.engine('eval')
            form = eng.describe()
            assert eng.form in ('x2m', 'fp16x2') and not eng.saturated()
        del model, acc
        torch.cuda.empty_cache()
    dp = (probs[None] - probs['fp32']).abs().max().item()
    d = np.abs(outs[None] - outs['fp32'])
    sure = np.abs(outs['fp32'][..., 0] - outs['fp32'][..., 1]) > 2
    print(f'C4 sub-volume 256^3 (27 blocks) in the default mode ({form["form"]}, calibration {form["calibration_max_abs_logit_diff_x2m_vs_fp16x2"]:.2e}): '
          f'max |blended probability diff| vs the fp32 mode = {dp:.2e}; max |uint8 diff| = {d.max()}, differing = {(d > 0).mean():.5f}; '
          f'class map compared on {sure.mean():.4f} of the voxels')
    assert dp <= 1e-3
    assert d.max() <= 1
    assert np.array_equal(outs[None].argmax(-1)[sure], outs['fp32'].argmax(-1)[sure])


@pytest.mark.parametrize('mode', ['bf16', 'compliant'])
def test_c4_eight_virtual_ranks_at_full_size(mode):
    """(mode 'compliant': the default prediction mode -- split precision with the cross terms on the fp8 matrix cores -- whose activations
    are 4 bytes per element instead of 2: the per-rank HBM footprint it reports is asserted)
    The 8-GPU geometry of C4 (1024^3, 1 331 blocks: 166 / 167 per rank, z-slabs of 128 planes) on ONE GPU: eight virtual ranks
    (threads, real device ops, the in-process communicator of test_gpu_shard.py) run shard.predict_volume_sharded in the bench's
    dtype (bf16).  Asserted: every rank's slab is byte-identical to the 1-rank result; the halo exchange delivers at most the
    rank's footprint (<= 3 block planes + overlap) instead of the whole volume; the probability pieces a rank sends stay within
    SURVEY 8e's estimate (<= 2 GB); the owners blend while the rounds run (>= 35 % of the pieces before the last round).  Reported: the summed 8-rank device time against
    the 1-rank time (no RCCL here: what it measures is the extra work of the sharded path -- halo copies, piece copies, cut
    blends, eight slab finalisations)."""
    import threading
    import time
    from interactive_unet import predict, shard
    from interactive_unet.unet import UNet
    from tests.test_gpu_shard import ThreadComm, _Shared
    S, C, V, world = 128, 2, (1024, 1024, 1024), 8
    p = unet_ref.init_params(dim=3, ncls=C, seed=5, randomize_bn=True)

    def model():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(num_classes=C, dim=3, act_dtype='bf16', infer_dtype=('bf16' if mode == 'bf16' else 'fp16x2'), pretrained=False)
        m.load_named(p)
        return m.cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(2)
    vol = torch.randint(0, 256, V, dtype=torch.uint8, device='cuda', generator=g)
    ops1 = shard.NativeOps(model(), C, S)
    assert getattr(ops1.eng, 'mixed', False) == (mode == 'compliant')
    times = []
    for _ in range(2):                                          # second run: caches warm
        torch.cuda.synchronize(); t0 = time.time()
        want, _ = shard.predict_volume_sharded(ops1, vol, V, S, 0.25)
        torch.cuda.synchronize(); times.append(time.time() - t0)
    t1 = times[-1]
    want = want.clone()
    del ops1
    torch.cuda.empty_cache()
    bounds, _ = shard.slab_bounds(V[0], world)
    shared = _Shared(world)
    opss = [shard.NativeOps(model(), C, S) for _ in range(world)]
    slabs = [vol[a:b].contiguous() for a, b in bounds]
    res, errs = [None] * world, []

    def run(r):
        try:
            res[r] = shard.predict_volume_sharded(opss[r], slabs[r], V, S, 0.25, comm=ThreadComm(shared, r))
        except Exception as e:                                   # pragma: no cover
            errs.append(e)
            shared.barrier.abort()
    t8 = None
    for _ in range(2):
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        torch.cuda.synchronize(); t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        torch.cuda.synchronize(); t8 = time.time() - t0
        assert not errs, errs
    nb = [r[1]['blocks'] for r in res]
    assert sum(nb) == 1331 and max(nb) - min(nb) <= 1
    for r, (a, b) in enumerate(bounds):
        out, st = res[r]
        assert torch.equal(out, want[a:b]), f'rank {r}: slab differs from the 1-rank result'
        f0, f1 = st['footprint']
        assert f1 - f0 <= 3 * 96 + 32 + 32 and st['halo_bytes_received'] <= (f1 - f0) * V[1] * V[2]
        assert st['bytes_sent'] <= 2.0e9, st
        # own blocks are blended as their forwards are issued; what waits for the last round is that round's blocks plus what
        # FOLLOWS them in flat order (the own tail's pieces, computed first for the next owner's sake, and the next rank's head)
        assert st['pieces_blended_before_last_round'] >= 0.35 * st['pieces_blended'], st
    sent = [r[1]['bytes_sent'] / 1e9 for r in res]
    halo = [r[1]['halo_bytes_received'] / 2 ** 20 for r in res]
    hbm = [r[1]['hbm_bytes'] / 2 ** 30 for r in res]
    # per-rank HBM: slab accumulators 128 x 1024^2 x (8 + 4 + 2) B = 1.75 GiB, <= 167 block probabilities x 16 MiB = 2.6 GiB, the receive
    # pool, the footprint window, and the engine's workspace for a batch of blocks (the compliant mode's is about twice the 16-bit one's)
    print(f'    per-rank HBM footprint ({mode}): {min(hbm):.2f}-{max(hbm):.2f} GiB of 288')
    assert max(hbm) <= (10.0 if mode == 'bf16' else 12.0), hbm
    print(f'C4 on 8 virtual ranks (one GPU, {mode}): blocks per rank {nb}; pieces sent per rank {min(sent):.2f}-{max(sent):.2f} GB; halo planes '
          f'received per rank {min(halo):.0f}-{max(halo):.0f} MiB (a whole-slab all-gather: 896 MiB); pieces blended before the last round '
          f'{min(r[1]["pieces_blended_before_last_round"] / r[1]["pieces_blended"] for r in res):.2f}+; summed 8-rank time {t8:.3f} s vs 1 rank '
          f'{t1:.3f} s: overhead {100 * (t8 / t1 - 1):.1f} %')
    assert t8 <= 1.25 * t1
