"""Benchmark of the native U-Net hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]         (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], "C3"): 3-D U-Net 4-level base 32, 1 -> 2 classes, bf16
activations, batches of 128^3 uint8 chunks resident in HBM.  One step = one training step
(forward + MCC+CE loss + backward + AdamW) on `--chunks` chunks, when the training path is
built, followed by one prediction pass (forward + softmax + Gaussian blend-accumulate +
normalise/quantise) over the same number of chunks.  value = voxels through the step / s,
whole job (all ranks; weak scaling: every rank owns its own chunks, gradients all-reduced
over RCCL when training is in the step).

The JSON line also carries
  roofline     -- the bottleneck 3x3x3 conv (dec0.conv1: 64 -> 32 channels at 128^3), timed
                  live with HIP events on the launch stream, against the dense MFMA peak;
  cpu_baseline -- the oracle (oracle/unet_ref.py, torch CPU fp32, all host cores) timed on a
                  bounded sample of the same workload, rank 0 / N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))

import numpy as np
import torch

MFMA_PEAK_TFLOPS = 2500.0          # dense bf16/fp16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0


def synth_chunks(n, S, seed, device):
    """Seeded, non-zero, smooth-ish uint8 chunks generated on the device (no files)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.rand((n, 1, S // 4, S // 4, S // 4), generator=g, device=device)
    x = torch.nn.functional.interpolate(x, size=(S, S, S), mode='trilinear', align_corners=False)
    x = x + 0.15 * torch.rand((n, 1, S, S, S), generator=g, device=device)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return (x * 254 + 1).to(torch.uint8).reshape(n, S, S, S)


def flops_per_voxel(dim, levels, base, cin, ncls):
    """Algorithmic forward FLOPs (2*MAC) per full-resolution voxel of the canonical U-Net."""
    ch = [base * 2 ** l for l in range(levels)]
    taps, f = 3 ** dim, 0.0
    for l in range(levels):
        f += 2 * taps * ((cin if l == 0 else ch[l - 1]) * ch[l] + ch[l] * ch[l]) / 2 ** (dim * l)
    for l in range(levels - 2, -1, -1):
        f += (2 * ch[l + 1] * ch[l] + 2 * taps * (2 * ch[l] * ch[l] + ch[l] * ch[l])) / 2 ** (dim * l)
    return f + 2 * ch[0] * ncls


def pmc_traffic():
    """HBM bytes per launch of the roofline kernel from the committed PMC profile (rocprofv3 cannot run
    inside the timed process): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_conv3_dec0conv1.json')) as f:
            return json.load(f)['traffic_bytes_per_launch']
    except Exception:
        return None


def conv_roofline(nv, dtype, S, iters=20):
    """dec0.conv1 of the 3-D net: Cin 64 -> Cout 32, 27 taps, one 128^3 chunk.  3 warm-up + 20 timed back-to-back launches,
    the same sequence as `tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 20` whose rocprofv3 summary is committed as
    profiles/r01_roofline_kernel_stats.csv (the launch time drifts from 0.22 to 0.28 ms over such a run as the clock
    settles under sustained MFMA load, so the number of launches matters)."""
    dev = 'cuda'
    cin, cout, taps = 64, 32, 27
    vox = S ** 3
    x = (torch.randn(cin * vox, device=dev) * 0.5).to(dtype)
    y = torch.empty(cout * vox, dtype=dtype, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.03
    dt = nv.DTYPE_CODE[dtype]
    lay = nv.lib().iunet_conv3_pick_layout(3, 1, S, S, S, cin, cout)
    pmode = 2 if lay > 0 else 0                 # layouts 1 and 2 share the K16 operator
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pmode), dtype=dtype, device=dev)
    bias = torch.zeros(cout, device=dev)
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pmode, nv.stream())
    run = lambda: nv.call('iunet_conv3_fwd', dt, 3, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wpk),
                          nv.ptr(bias), None, 1, S, S, S, cin, cout, 2, lay, nv.stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * taps * cin * cout * vox
    ach = flops / (ms * 1e-3) / 1e12
    kname = {0: 'conv3_mfma_kernel', 1: 'conv3_v2_kernel', 2: 'conv3_v4_kernel'}[lay]
    return {'bound': 'mfma', 'kernel': kname + f'<{"bf16" if dtype == torch.bfloat16 else "f16"},3> (dec0.conv1 64->32 @128^3)',
            'achieved': round(ach, 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_PEAK_TFLOPS, 4),
            'ms_per_launch': round(ms, 4), 'flops_per_launch': flops, 'traffic': pmc_traffic()}


def cpu_baseline(ncls):
    """Oracle (torch CPU fp32) on the host cores: bounded sample of the same step --
    one training step (forward, MCC+CE loss, autograd backward, AdamW) plus one prediction
    forward, on ONE 64^3 chunk (1/8 of a bench chunk), repeated twice."""
    from oracle import unet_ref, metrics_ref
    cores = min(16, len(os.sched_getaffinity(0)))        # a 1-GPU box gets a 16-core share
    torch.set_num_threads(cores)
    Sc = 64
    p = unet_ref.init_params(dim=3, ncls=ncls, seed=0)
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in pr.items()}
    v = {k: torch.zeros_like(v) for k, v in pr.items()}
    x = torch.rand(1, 1, Sc, Sc, Sc)
    lab = (x > 0.5)
    y = torch.cat([~lab, lab], 1).float()
    from interactive_unet import metrics as host_metrics     # plain torch ops on the host (differentiable)
    reps, t0 = 2, None
    for it in range(reps + 1):
        if it == 1:
            t0 = time.time()                                   # first iteration warms the thread pool
        probs = unet_ref.forward(pr, x, dim=3, training=True)
        loss = host_metrics.mcc_ce_loss(probs, y, None, axes=[0, 2, 3, 4])
        grads = torch.autograd.grad(loss, [t for k, t in pr.items() if t.requires_grad])
        with torch.no_grad():
            g = dict(zip([k for k, t in pr.items() if t.requires_grad], grads))
            unet_ref.adamw_step({k: t.data for k, t in pr.items()}, g, m, v, it + 1, 1e-4)
            unet_ref.forward(pr, x, dim=3)
    dt = time.time() - t0
    return {'value': round(reps * 2 * Sc ** 3 / dt, 1), 'unit': 'voxels/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} x (1 training step + 1 prediction forward) of the fp32 oracle 3-D U-Net on one '
                      f'{Sc}^3 chunk (1/8 bench chunk), torch CPU, {cores} threads; voxels counted once per leg'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--chunks', type=int, default=2, help='128^3 chunks per GPU per step')
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from interactive_unet import _native as nv

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    S, B, ncls = args.size, args.chunks, 2
    dev = torch.device('cuda', local)
    import warnings
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(lr=1e-4, num_classes=ncls, dim=3, act_dtype=args.dtype, pretrained=False)
    model.reset_parameters(seed=0)                                      # random-init weights of the canonical architecture
    model = model.to(dev)
    trainer = TrainEngine(model, lr=1e-4, loss_kind='mcc_ce', process_group=(dist.group.WORLD if dist else None))
    chunks = synth_chunks(B, S, 1234 + rank, dev)
    X = chunks.reshape(B, 1, S, S, S)                                   # uint8, /255 in the first conv
    lab = X > 127
    y = torch.cat([~lab, lab], 1).to(torch.float16)                     # loader contract: fp16 one-hot (loader.py:150-152)
    w = (torch.rand((B, 1, S, S, S), device=dev) > 0.1).to(torch.float16).expand(B, ncls, S, S, S).contiguous()
    y = y * w
    # predict leg: ONE volume shared by all ranks, Z = 96*B*world + 32 planes of 128 x 128 -> exactly B
    # overlapping 128^3 blocks per rank (predict.py:362-411 grid, overlap 0.25); every rank owns a z-slab,
    # slabs are all-gathered and the overlapping accumulator planes exchanged over RCCL (shard.py)
    from interactive_unet import shard
    stride = int(S * 0.75)
    V = (stride * B * world + (S - stride), S, S)
    bounds, _ = shard.slab_bounds(V[0], world)
    my_slab = synth_chunks(1, S, 4321 + rank, dev).reshape(S, S, S)
    my_slab = my_slab.repeat(-(-(bounds[rank][1] - bounds[rank][0]) // S), 1, 1)[:bounds[rank][1] - bounds[rank][0]].contiguous()
    ops = shard.NativeOps(model, ncls, S)
    legs = {'train': 0.0, 'predict': 0.0}
    info = {}

    def step(timed=False):
        t0 = time.time()
        trainer.train_step(X, y, w, sync=False)
        if timed:
            torch.cuda.synchronize(); t1 = time.time(); legs['train'] += t1 - t0
        model.engine('eval')                                            # re-packs the updated weights (BN folded)
        out_u8, st = shard.predict_volume_sharded(ops, my_slab, V, S, 0.25, group=(dist.group.WORLD if dist else None))
        info.update(st)
        if timed:
            torch.cuda.synchronize(); legs['predict'] += time.time() - t1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # leg split, measured in separate (untimed-for-value) steps so the timed region has no extra syncs
    for _ in range(2):
        step(timed=True)
    vox_per_step = 2 * B * S ** 3 * world            # every chunk voxel goes through the train leg and the predict leg
    value = vox_per_step * args.steps / dt

    out = None
    if rank == 0:
        roof = conv_roofline(nv, dtype, S)
        out = {
            'metric': 'voxels/sec (train step + full-volume predict) on 128^3 chunks',
            'value': round(value, 1), 'unit': 'voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'C3: 3-D U-Net 4-level base 32, 1->{ncls} classes, {B} x {S}^3 uint8 chunks per GPU '
                                   f'per step; step = 1 training step (forward with BatchNorm batch stats, MCC+CE loss, '
                                   f'backward, AdamW, weight re-pack; gradients all-reduced over RCCL for N > 1) on the '
                                   f'{B} chunks + tiled prediction of one {V[0]}x{S}x{S} uint8 volume shared by all ranks = '
                                   f'{B} overlapping {S}^3 blocks per GPU (reflect-padded block gather, forward + softmax, '
                                   f'Gaussian blend-accumulate, slab all-gather + overlap exchange over RCCL for N > 1, '
                                   f'normalise/quantise); voxels = chunk voxels, counted once per leg',
                       'predict_volume': list(V), 'predict_blocks_per_gpu': info.get('blocks'),
                       'predict_exchange_bytes_sent_rank0': info.get('bytes_sent'),
                       'chunks_per_gpu': B, 'chunk': S, 'levels': 4, 'base': 32,
                       'fwd_flop_per_voxel': flops_per_voxel(3, 4, 32, 1, ncls)},
            'roofline': roof,
        }
        fpv = flops_per_voxel(3, 4, 32, 1, ncls)
        out['legs'] = {'train_ms': round(legs['train'] / 2 * 1e3, 3), 'predict_ms': round(legs['predict'] / 2 * 1e3, 3),
                       'train_voxels_per_s_per_gpu': round(B * S ** 3 / (legs['train'] / 2), 1),
                       'predict_voxels_per_s_per_gpu': round(B * S ** 3 / (legs['predict'] / 2), 1),
                       'train_tflops_per_gpu(3x fwd)': round(3 * fpv * B * S ** 3 / (legs['train'] / 2) / 1e12, 1),
                       'predict_tflops_per_gpu': round(fpv * B * S ** 3 / (legs['predict'] / 2) / 1e12, 1)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(ncls)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
