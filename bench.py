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


def conv_roofline(nv, dtype, S, iters=10):
    """dec0.conv1 of the 3-D net: Cin 64 -> Cout 32, 27 taps, one 128^3 chunk."""
    dev = 'cuda'
    cin, cout, taps = 64, 32, 27
    vox = S ** 3
    x = (torch.randn(cin * vox, device=dev) * 0.5).to(dtype)
    y = torch.empty(cout * vox, dtype=dtype, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.03
    wpk = torch.empty(cout * cin * taps, dtype=dtype, device=dev)
    bias = torch.zeros(cout, device=dev)
    dt = nv.DTYPE_CODE[dtype]
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, 0, nv.stream())
    run = lambda: nv.call('iunet_conv3_fwd', dt, 3, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wpk),
                          nv.ptr(bias), None, 1, S, S, S, cin, cout, 2, nv.stream())
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
    return {'bound': 'mfma', 'kernel': 'conv3_mfma_kernel<bf16,3,2> (dec0.conv1 64->32 @128^3)',
            'achieved': round(ach, 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_PEAK_TFLOPS, 4),
            'ms_per_launch': round(ms, 4), 'flops_per_launch': flops, 'traffic': None}


def cpu_baseline(S, ncls):
    """Oracle forward on the host cores: bounded sample of the same workload."""
    from oracle import unet_ref
    torch.set_num_threads(os.cpu_count())
    p = unet_ref.init_params(dim=3, ncls=ncls, seed=0)
    x = torch.rand(1, 1, S, S, S)
    with torch.inference_mode():
        unet_ref.forward(p, x[:, :, :32, :32, :32], dim=3)            # warm the thread pool
        t0 = time.time()
        reps = 0
        while reps < 2:
            unet_ref.forward(p, x, dim=3)
            reps += 1
        dt = time.time() - t0
    return {'value': round(reps * S ** 3 / dt, 1), 'unit': 'voxels/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': f'{reps} forward passes (predict leg only) of the fp32 oracle 3-D U-Net on one {S}^3 chunk, '
                      f'torch CPU, {os.cpu_count()} threads'}


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
    from interactive_unet.engine import Engine
    from interactive_unet import predict as npredict
    from oracle import unet_ref                      # weights only (init), not in the timed path

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    S, B, ncls = args.size, args.chunks, 2
    dev = torch.device('cuda', local)
    params = {k: v.to(dev) for k, v in unet_ref.init_params(dim=3, ncls=ncls, seed=0).items()}
    eng = Engine(dim=3, ncls=ncls, act_dtype=dtype, device=dev)
    eng.load_eval(params)
    chunks = synth_chunks(B, S, 1234 + rank, dev)
    acc = npredict.VolumeAccumulator((B * S, S, S), ncls, S, dev)      # chunks stacked along z

    def step():
        eng_probs = acc.block_probs
        for b in range(B):
            eng.infer(chunks[b], (S ** 3, S ** 3, S * S, S, 1), 1, S, S, S, probs=eng_probs,
                      out_strides=(0, 1, S * S * ncls, S * ncls, ncls))
            acc.blend((b * S, 0, 0, (b + 1) * S, S, S), (0, 0, 0, S, S, S))
        acc.finalize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        acc.reset()
        step()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        acc.reset()
        step()
    barrier()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    vox_per_step = B * S ** 3 * world
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
                                   f'per step; step = predict leg only (forward + softmax + Gaussian blend + '
                                   f'normalise/quantise); training leg not in the step yet',
                       'chunks_per_gpu': B, 'chunk': S, 'levels': 4, 'base': 32,
                       'fwd_flop_per_voxel': unet_ref.flops_per_voxel(3, 4, 32, 1, ncls)},
            'roofline': roof,
        }
        fwd_tflops = unet_ref.flops_per_voxel(3, 4, 32, 1, ncls) * B * S ** 3 * args.steps / dt / 1e12
        out['predict_tflops_per_gpu'] = round(fwd_tflops, 1)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(S, ncls)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
