"""Benchmark of the native U-Net hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c4|c5|c2]

`--gpus N` with N > 1 and no torch.distributed environment: bench.py starts itself under
`python -m torch.distributed.run --nproc-per-node N` (child processes; this process has not touched the GPU) and
exits with their code.  Under a launcher (WORLD_SIZE set) it is one rank per GPU over RCCL.

Workloads (BASELINE.json `configs`; the metric is "voxels/sec (train step + full-volume predict) on 128^3 chunks"):

  c3 (default, configs[2])  3-D U-Net 4-level base 32, 1 -> 2 classes, bf16, `--chunks` 128^3 uint8 chunks per GPU
       resident in HBM.  One step = one training step on the chunks (forward with BatchNorm batch statistics, MCC+CE
       loss, backward, AdamW, weight re-pack; gradients all-reduced over RCCL for N > 1) + tiled prediction of one
       (96*chunks*N + 32) x 128 x 128 volume shared by the N ranks = `chunks` overlapping 128^3 blocks per GPU
       (reflect-padded block gather, forward + softmax, Gaussian blend, piece exchange for N > 1, normalise +
       quantise).  value = (training chunk voxels + UNIQUE predicted volume voxels) / s, whole job, weak scaling.
       The same run then times the C4 volume once (`"c4"` object of the line; `--c4-reps 0` skips it).
  c4 (configs[3])  the fixed 1024^3 uint8 volume (generated on the device from the voxel coordinates, identical for
       every N), 1 331 blocks of 128^3, overlap 0.25, sharded over the N ranks (shard.py): one step = one whole-volume
       prediction, wall time including every exchange.  value = unique volume voxels / s; STRONG scaling.
  c5 (configs[4])  3-D 5-level base 64, 4 classes: training step in bf16 + prediction with e4m3 weights, 1 chunk.
  c2 (configs[1])  2-D 4-level base 32, fp16: training step on 8 slices of 512^2 + prediction of the 8 slices.

`value` / `ms_per_step` are the COMPLIANT pairing: training in the workload's 16-bit dtype (the reference trains under '16-mixed',
trainer.py:59) + prediction in the default, tolerance-meeting mode (split precision; the form -- x2m or fp16x2 -- is selected by
engine_auto.EngineAuto's calibration on the model's own weights and reported in `config.predict_dtype` /
`config.predict_mode_selection`; the reference predicts in fp32, predict.py:30-35).

The JSON line also carries
  roofline     -- the workload's bottleneck 3x3(x3) conv (dec0.conv1), timed live with HIP events on the launch
                  stream over 50 back-to-back launches (average = `achieved`: the figure the committed rocprofv3
                  summary of the same launch sequence reproduces; the first 8 and the last 20 are given as burst /
                  settled: the clock drops under sustained MFMA load), against the dense MFMA peak; `in_situ` = the
                  same layer's launches inside real steps (events around each); `predict_kernel` = the prediction leg's
                  split-precision conv of that layer inside real steps;
  throughput_mode -- the same step timed a second time with the prediction leg in the 16-bit dtype (its deviation from the
                  fp32 path is in `parity`);
  parity       -- max |logit| deviation and class-map mismatches of the benchmarked dtype, fp16x2 and x2m against the native fp32
                  parity mode on one chunk, each with the population of its tie band (voxels whose fp32 top-2 margin is within twice
                  the measured error) and the mismatches OUTSIDE it (must be 0); plus, when the CPU baseline runs, every mode and
                  the default mode's own selection against the CPU fp32 oracle on its sample;
  cpu_baseline -- the oracle (oracle/unet_ref.py, torch CPU fp32, host cores) timed on a bounded sample of the same
                  workload, rank 0 / N = 1 only;
  legs.predict_2p5d -- the reference's own predict semantics (predict.py:79-112): one 128^3 block through the
                  2-D net along 3 axes, in the default mode and fp16, with the oracle's predict_block on the host cores beside it;
  rccl_ranks   -- N > 1: the result of an all-reduce of ones over the process group (must equal n_gpus).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))

import numpy as np
import torch

MFMA_PEAK_TFLOPS = 2500.0          # dense bf16/fp16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    #        dim levels base ncls dtype   infer weights  tile            chunks
    'c3': dict(dim=3, levels=4, base=32, ncls=2, dtype='bf16', wq=None, tile=(128, 128, 128), chunks=2),
    'c4': dict(dim=3, levels=4, base=32, ncls=2, dtype='bf16', wq=None, tile=(128, 128, 128), chunks=0),
    'c5': dict(dim=3, levels=5, base=64, ncls=4, dtype='bf16', wq='fp8_e4m3', tile=(128, 128, 128), chunks=1),
    'c2': dict(dim=2, levels=4, base=32, ncls=2, dtype='f16', wq=None, tile=(512, 512), chunks=8),
}


def synth_chunks(n, shape, seed, device):
    """Seeded, non-zero, smooth-ish uint8 tiles generated on the device (no files)."""
    g = torch.Generator(device=device).manual_seed(seed)
    nd = len(shape)
    x = torch.rand((n, 1) + tuple(s // 4 for s in shape), generator=g, device=device)
    x = torch.nn.functional.interpolate(x, size=shape, mode='trilinear' if nd == 3 else 'bilinear', align_corners=False)
    x = x + 0.15 * torch.rand((n, 1) + tuple(shape), generator=g, device=device)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return (x * 254 + 1).to(torch.uint8).reshape((n,) + tuple(shape))


def synth_volume_slab(z0, z1, Y, X, device):
    """Planes [z0, z1) of the C4 volume: a function of the voxel coordinates only (smooth blobs + a coordinate hash for
    texture), so every rank of every world size generates the same 1024^3 volume without a file."""
    out = torch.empty((z1 - z0, Y, X), dtype=torch.uint8, device=device)
    yy = torch.arange(Y, device=device, dtype=torch.float32).view(1, Y, 1)
    xx = torch.arange(X, device=device, dtype=torch.float32).view(1, 1, X)
    yi = torch.arange(Y, device=device, dtype=torch.int64).view(1, Y, 1)
    xi = torch.arange(X, device=device, dtype=torch.int64).view(1, 1, X)
    for a in range(z0, z1, 32):
        b = min(a + 32, z1)
        zz = torch.arange(a, b, device=device, dtype=torch.float32).view(-1, 1, 1)
        zi = torch.arange(a, b, device=device, dtype=torch.int64).view(-1, 1, 1)
        f = torch.sin(0.051 * zz + 0.3) + torch.sin(0.043 * yy + 1.0) * torch.cos(0.037 * xx) + torch.sin(0.029 * (xx + zz) + 2.0)
        h = ((zi * 73856093) ^ (yi * 19349663) ^ (xi * 83492791)) & 31
        out[a - z0:b - z0] = (128.0 + 30.0 * f + h.float()).clamp_(1, 255).to(torch.uint8)
    return out


def flops_per_voxel(dim, levels, base, cin, ncls):
    """Algorithmic forward FLOPs (2*MAC) per full-resolution voxel of the canonical U-Net."""
    ch = [base * 2 ** l for l in range(levels)]
    taps, f = 3 ** dim, 0.0
    for l in range(levels):
        f += 2 * taps * ((cin if l == 0 else ch[l - 1]) * ch[l] + ch[l] * ch[l]) / 2 ** (dim * l)
    for l in range(levels - 2, -1, -1):
        f += (2 * ch[l + 1] * ch[l] + 2 * taps * (2 * ch[l] * ch[l] + ch[l] * ch[l])) / 2 ** (dim * l)
    return f + 2 * ch[0] * ncls


def pmc_traffic(workload, tiles=1):
    """HBM bytes per launch of the roofline kernel from the committed PMC profile (rocprofv3 cannot run inside the timed
    process): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes, scaled to the tiles of the launch.  None where
    no profile is committed."""
    for name in (f'r05_pmc_{workload}_dec0conv1.json', 'r05_pmc_c3_dec0conv1.json' if workload == 'c4' else '',
                 f'r04_pmc_{workload}_dec0conv1.json', 'r04_pmc_c3_dec0conv1.json' if workload == 'c4' else '',
                 f'r03_pmc_{workload}_dec0conv1.json', 'r03_pmc_c3_dec0conv1.json' if workload == 'c4' else ''):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                d = json.load(f)
            return int(d['traffic_bytes_per_launch'] * tiles / d.get('tiles_per_launch', 1))
        except Exception:
            continue
    return None


FP8_PEAK_TFLOPS = 5000.0        # dense e4m3 peak of the K = 128 matrix instructions (MI355X_MICROARCH.md, "Matrix cores")


def _roof_fields(flops, alg_bytes, ms, burst=None, settled=None, MFMA_PEAK_TFLOPS=MFMA_PEAK_TFLOPS):
    """Roofline numbers of one launch shape: bound by arithmetic intensity (activations once in, once out) against the ridge
    of the two peaks; `ms` = average launch duration.  MFMA_PEAK_TFLOPS: the dense peak of the instruction the kernel issues."""
    tf = lambda t: flops / (t * 1e-3) / 1e12
    gbs = lambda t: alg_bytes / (t * 1e-3) / 1e9
    hbm_bound = flops / alg_bytes < MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    out = {'bound': 'hbm' if hbm_bound else 'mfma'}
    if hbm_bound:
        out.update({'achieved': round(gbs(ms), 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs(ms) / HBM_PEAK_GBS, 4),
                    'tflops': round(tf(ms), 2), 'mfma_frac': round(tf(ms) / MFMA_PEAK_TFLOPS, 4)})
        rate, peak = gbs, HBM_PEAK_GBS
    else:
        out.update({'achieved': round(tf(ms), 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf(ms) / MFMA_PEAK_TFLOPS, 4),
                    'hbm_gbs_algorithmic': round(gbs(ms), 1)})
        rate, peak = tf, MFMA_PEAK_TFLOPS
    if burst is not None:
        out.update({'burst_ms_first8': round(burst, 4), 'burst_frac': round(rate(burst) / peak, 4),
                    'settled_ms_last20': round(settled, 4), 'settled_frac': round(rate(settled) / peak, 4)})
    out.update({'ms_per_launch': round(ms, 4), 'flops_per_launch': flops, 'algorithmic_bytes_per_launch': alg_bytes,
                'flop_per_byte': round(flops / alg_bytes, 1)})
    return out


def _roof_kernel_name(nv, cfg, dtype, N):
    dim, base = cfg['dim'], cfg['base']
    shape = cfg['tile']
    D, H, W = shape if dim == 3 else (1,) + tuple(shape)
    lay = nv.lib().iunet_conv3_pick_layout(dim, N, D, H, W, 2 * base, base)
    if nv.lib().iunet_conv3_compact_ok(dim, N, D, H, W, 2 * base, base, 0, 0):
        lay = 3                                    # what the engines launch (PackedConv.pick): the compact operator, padding-free step
    k128 = bool(cfg['wq']) and dim == 3 and nv.lib().iunet_f8_pack_order(27, 2 * base) == 1
    kname = 'conv3_f8k_kernel' if k128 else 'conv3_f8_kernel' if cfg['wq'] else {0: 'conv3_mfma_kernel', 1: 'conv3_v2_kernel', 2: 'conv3_v4_kernel', 3: 'conv3_v4_kernel'}[lay]
    tname = 'bf16' if dtype == torch.bfloat16 else 'f16'
    if k128:
        return (f'{kname}<{tname}> (dec0.conv1 {2 * base}->{base} @ {N} x {"x".join(str(s) for s in shape)}, v_mfma_f32_16x16x128_f8f6f4, '
                f'e4m3 activation planes in and out)'), lay
    return f'{kname}<{tname},{dim}> (dec0.conv1 {2 * base}->{base} @ {N} x {"x".join(str(s) for s in shape)})', lay


def _f8_k128(nv, cfg):
    return bool(cfg['wq']) and cfg['dim'] == 3 and nv.lib().iunet_f8_pack_order(27, 2 * cfg['base']) == 1


def conv_roofline_in_situ(nv, cfg, workload, dtype, events):
    """The workload's bottleneck conv (dec0.conv1) as the STEP launches it: HIP events recorded around every forward launch
    of that layer inside real steps (engine / train_engine `probe`), on the stream the kernels run on.  achieved = algorithmic
    FLOPs (or bytes) of those launches / their summed duration."""
    dim, base = cfg['dim'], cfg['base']
    cin, cout, taps = 2 * base, base, 3 ** dim
    vox = int(np.prod(cfg['tile']))
    torch.cuda.synchronize()
    per = [(e0.elapsed_time(e1), n) for e0, e1, n in events]
    ms_tot = sum(t for t, _ in per)
    n_tot = sum(n for _, n in per)
    N = max(n for _, n in per)
    ms = ms_tot / len(per) * (N * len(per) / n_tot)             # average duration of a launch of N tiles
    flops = 2.0 * taps * cin * cout * vox * N
    k128 = _f8_k128(nv, cfg)            # e4m3 planes in and out: one byte per element; priced against the fp8 dense peak
    out = _roof_fields(flops, (cin + cout) * (1.0 if k128 else 2.0) * vox * N, ms, MFMA_PEAK_TFLOPS=FP8_PEAK_TFLOPS if k128 else MFMA_PEAK_TFLOPS)
    name, _ = _roof_kernel_name(nv, cfg, dtype, N)
    out = {'bound': out.pop('bound'), 'kernel': name, **out}
    out.update({'launches': len(per), 'tiles_per_launch': N, 'measured': 'in situ: HIP events around the forward launches of this layer '
                'inside real steps (prediction forward; for the 16-bit workloads also the training forward with its BatchNorm-statistics '
                'epilogue)',
                'min_ms': round(min(t for t, _ in per), 4), 'max_ms': round(max(t for t, _ in per), 4),
                'traffic': pmc_traffic(workload, N)})
    return out


def conv_roofline_back_to_back(nv, cfg, workload, dtype, N, iters=50):
    """The same layer alone: 3 warm-up + `iters` back-to-back launches, an event around each (what `rocprofv3 --kernel-trace
    --stats` of tools/bench_conv.py averages too).  Under sustained MFMA load the chip lowers its clock, so this reads lower
    than the in-situ figure: burst = first 8 launches, settled = last 20."""
    dev = 'cuda'
    dim, base = cfg['dim'], cfg['base']
    cin, cout, taps = 2 * base, base, 3 ** dim
    shape = cfg['tile']
    D, H, W = shape if dim == 3 else (1,) + tuple(shape)
    vox = D * H * W
    x = (torch.randn(N * cin * vox, device=dev) * 0.5).to(dtype)
    y = torch.empty(N * cout * vox, dtype=dtype, device=dev)
    w = torch.randn((cout, cin) + (3,) * dim, device=dev) * 0.03
    dt = nv.DTYPE_CODE[dtype]
    name, lay = _roof_kernel_name(nv, cfg, dtype, N)
    pmode = 6 if lay == 3 else 2 if lay > 0 else 0          # layouts 1 and 2 share the K16 operator, layout 3 takes the compact one
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, taps, pmode), dtype=dtype, device=dev)
    bias = torch.zeros(cout, device=dev)
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, taps, pmode, nv.stream())
    run = lambda: nv.call('iunet_conv3_fwd', dt, dim, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wpk),
                          nv.ptr(bias), None, N, D, H, W, cin, cout, 2, lay, nv.stream())
    if cfg['wq']:            # C5: the same layer on the fp8 matrix cores (e4m3 weight bytes, activations rounded in the loaders)
        wb = torch.zeros(nv.lib().iunet_f8_pack_conv3_bytes(cout, cin, taps), dtype=torch.uint8, device=dev)
        wsc = torch.empty(cout, device=dev)
        nv.call('iunet_f8_pack_conv3', nv.ptr(w), None, None, None, None, 1e-5, nv.ptr(wb), nv.ptr(wsc), None, cout, cin, taps, nv.stream())
        run = lambda: nv.call('iunet_conv3_f8_fwd', dt, dim, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wb), nv.ptr(wsc),
                              nv.ptr(bias), N, D, H, W, cin, cout, 2, None, nv.stream())
        if _f8_k128(nv, cfg):    # as the engine launches it: e4m3 activation planes in and out (one byte per element)
            xq = (torch.randn(N * cin * vox, device=dev) * 2).to(torch.float8_e4m3fn).view(torch.uint8)
            yq = torch.empty(N * cout * vox, dtype=torch.uint8, device=dev)
            run = lambda: nv.call('iunet_conv3_f8_fwd_q', dt, dim, nv.ptr(xq), cin * vox, 1, nv.ptr(yq), cout * vox, 1, nv.ptr(wb), nv.ptr(wsc),
                                  nv.ptr(bias), N, D, H, W, cin, cout, 2, None, nv.stream())
    for _ in range(3):
        run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    ms = ev[0].elapsed_time(ev[iters]) / iters
    k128 = _f8_k128(nv, cfg)
    out = _roof_fields(2.0 * taps * cin * cout * vox * N, (cin + cout) * (1.0 if k128 else 2.0) * vox * N, ms, sum(per[:8]) / 8, sum(per[-20:]) / 20,
                       MFMA_PEAK_TFLOPS=FP8_PEAK_TFLOPS if k128 else MFMA_PEAK_TFLOPS)
    out = {'bound': out.pop('bound'), 'kernel': name, **out}
    out.update({'launches': iters, 'tiles_per_launch': N, 'measured': f'3 warm-up + {iters} back-to-back launches of the layer alone'})
    return out


def native_parity(model, cfg, chunk_u8):
    """The benchmarked dtype and the split-precision mode (fp16x2) against the native fp32 parity mode on one tile (same weights):
    max |logit| difference and class-map mismatches.  Device only -- the fp32 and fp16x2 modes are themselves checked against the
    CPU oracle in the tests."""
    from interactive_unet.engine_f32 import EngineF32
    from interactive_unet.engine_x2 import EngineX2
    dim, ncls = cfg['dim'], cfg['ncls']
    shape = tuple(chunk_u8.shape)
    D, H, W = shape if dim == 3 else (1,) + shape
    vox = D * H * W
    norm = cfg.get('norm', 'batch')
    e32 = EngineF32(dim, cfg['levels'], cfg['base'], 1, ncls, model.device, norm=norm)
    e32.load_eval(model.named_tensors())
    engs = [model.engine('eval'), e32]
    names = []
    if not cfg['wq']:
        for nm, mixed in (('fp16x2', False), ('x2m', True)):
            ex2 = EngineX2(dim, cfg['levels'], cfg['base'], 1, ncls, model.device, mixed=mixed, norm=norm)
            ex2.load_eval(model.named_tensors())
            engs.append(ex2)
            names.append(nm)
    outs = []
    for e in engs:
        lg = torch.empty((1, ncls) + shape, device=model.device)
        cl = torch.empty((1, vox), dtype=torch.uint8, device=model.device)
        e.infer(chunk_u8, (vox, vox, H * W, W, 1), 1, D, H, W, logits=lg, cls=cl)
        outs.append((lg, cl))
    torch.cuda.synchronize()
    (lg, cl), (lg32, cl32) = outs[:2]
    top2 = torch.topk(lg32, 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).reshape(1, vox)                 # top-2 margin of the fp32 mode: a mismatch inside 2 x err is a tie

    def band(err, mism):
        """(voxels whose fp32 margin is <= 2 err: the tie band, mismatches outside it: must be 0)"""
        return int((margin <= 2 * err).sum()), int((mism & (margin > 2 * err)).sum())
    err = float((lg - lg32).abs().max())
    res = {'tile': list(shape), 'dtype_vs': 'native fp32 parity mode (engine_f32, f32-input MFMA)',
           'max_abs_logit_vs_fp32': err, 'logit_scale': float(lg32.abs().max()),
           'argmax_mismatch': int((cl != cl32).sum()), 'voxels': vox}
    res['tie_band_voxels'], res['argmax_mismatch_outside_tie_band'] = band(err, cl != cl32)
    for nm, (lgx, clx) in zip(names, outs[2:]):
        e = float((lgx - lg32).abs().max())
        res[f'{nm}_max_abs_logit_vs_fp32'] = e
        res[f'{nm}_argmax_mismatch'] = int((clx != cl32).sum())
        res[f'{nm}_tie_band_voxels'], res[f'{nm}_argmax_mismatch_outside_tie_band'] = band(e, clx != cl32)
    del engs
    return res, lg32


def cpu_baseline(cfg, model=None, chunk_u8=None):
    """Oracle (torch CPU fp32) on the host cores: bounded sample of the same step -- one training step (forward,
    MCC+CE loss, autograd backward, AdamW) plus one prediction forward on ONE reduced tile, repeated twice.  With
    `model` given, the oracle's logits of the sample are also held against the native modes (checker use)."""
    from oracle import unet_ref
    from interactive_unet import metrics as host_metrics     # plain torch ops on the host (differentiable)
    cores = min(16, len(os.sched_getaffinity(0)))        # a 1-GPU box gets a 16-core share
    torch.set_num_threads(cores)
    dim, levels, base, ncls = cfg['dim'], cfg['levels'], cfg['base'], cfg['ncls']
    # 2-D: the whole C1 case of BASELINE.json configs[0] (one 512 x 512 slice: training step + validation forward = the
    # reference's "one epoch" on its CPU-runnable configuration); 3-D: a reduced tile
    shp = (64, 64, 64) if dim == 3 and base == 32 else ((32, 32, 32) if dim == 3 else (512, 512))
    frac = float(np.prod(shp)) / float(np.prod(cfg['tile']))
    p = unet_ref.init_params(dim=dim, levels=levels, base=base, ncls=ncls, seed=0)
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in pr.items()}
    v = {k: torch.zeros_like(v) for k, v in pr.items()}
    x = torch.rand((1, 1) + shp)
    lab = (x * ncls).long().clamp_(max=ncls - 1)
    y = torch.cat([(lab == c) for c in range(ncls)], 1).float()
    axes = [0] + list(range(2, 2 + dim))
    reps, t0 = 2, None
    for it in range(reps + 1):
        if it == 1:
            t0 = time.time()                                   # first iteration warms the thread pool
        probs = unet_ref.forward(pr, x, dim=dim, levels=levels, training=True, norm=cfg.get('norm', 'batch'))
        loss = host_metrics.mcc_ce_loss(probs, y, None, axes=axes)
        grads = torch.autograd.grad(loss, [t for k, t in pr.items() if t.requires_grad])
        with torch.no_grad():
            g = dict(zip([k for k, t in pr.items() if t.requires_grad], grads))
            unet_ref.adamw_step({k: t.data for k, t in pr.items()}, g, m, v, it + 1, 1e-4)
            unet_ref.forward(pr, x, dim=dim, levels=levels, norm=cfg.get('norm', 'batch'))
    dt = time.time() - t0
    nvox = int(np.prod(shp))
    out = {'value': round(reps * 2 * nvox / dt, 1), 'unit': 'voxels/s', 'cores': cores, 'kind': 'port',
           'sample': f'{reps} x (1 training step + 1 prediction forward) of the fp32 oracle ({dim}-D U-Net, {levels} levels, '
                     f'base {base}) on one {"x".join(map(str, shp))} tile ({frac:.3g} of a bench tile), torch CPU, '
                     f'{cores} threads; voxels counted once per leg'}
    if dim == 3 and base == 32 and tuple(cfg['tile']) == (128, 128, 128):
        # the prediction forward on ONE full 128^3 chunk (it fits the host: ~7 s on 16 threads)
        x128 = torch.rand((1, 1, 128, 128, 128))
        with torch.no_grad():
            t1 = time.time()
            unet_ref.forward(pr, x128, dim=dim, levels=levels, norm=cfg.get('norm', 'batch'))
            d128 = time.time() - t1
        out['forward_128'] = {'seconds': round(d128, 2), 'voxels_per_s': round(128 ** 3 / d128, 1),
                              'sample': 'one prediction forward of the fp32 oracle on a full 128x128x128 chunk (BASELINE.md B3)'}
        del x128
    chk = None
    if model is not None and chunk_u8 is not None:
        # checker: oracle logits of a crop of the bench tile with the MODEL's current weights vs both native modes
        from interactive_unet.engine_f32 import EngineF32
        sl = tuple(slice(0, s) for s in shp)
        crop = chunk_u8[sl].contiguous()
        params = {k: t.detach().float().cpu() for k, t in model.named_tensors().items()}
        with torch.no_grad():
            ref = unet_ref.forward_logits(params, crop.cpu().float()[None, None] / 255.0, dim=dim, levels=levels, norm=cfg.get('norm', 'batch'))
        D, H, W = shp if dim == 3 else (1,) + shp
        norm = cfg.get('norm', 'batch')
        e32 = EngineF32(dim, levels, base, 1, ncls, model.device, norm=norm)
        e32.load_eval(model.named_tensors())
        chk = {'sample': f'{"x".join(map(str, shp))} crop of a bench tile, current weights'}
        modes = [(cfg['dtype'], model.engine('eval')), ('fp32_mode', e32)]
        if not cfg['wq']:
            from interactive_unet.engine_auto import EngineAuto
            from interactive_unet.engine_x2 import EngineX2
            ea = EngineAuto(dim, levels, base, 1, ncls, model.device, norm=norm)        # what UNet() predicts in: the calibrated choice
            ea.load_eval(model.named_tensors())
            modes.append(('default_mode', ea))
            for nm, mixed in (('fp16x2', False), ('x2m', True)):
                ex2 = EngineX2(dim, levels, base, 1, ncls, model.device, mixed=mixed, norm=norm)
                ex2.load_eval(model.named_tensors())
                modes.append((nm, ex2))
        for name, e in modes:
            lg = torch.empty((1, ncls) + shp, device=model.device)
            cl = torch.empty((1, nvox), dtype=torch.uint8, device=model.device)
            e.infer(crop, (nvox, nvox, H * W, W, 1), 1, D, H, W, logits=lg, cls=cl)
            torch.cuda.synchronize()
            err = float((lg.cpu() - ref).abs().max())
            mism = cl.cpu().long().reshape(-1) != ref.argmax(1).reshape(-1)
            chk[f'{name}_max_abs_logit_vs_cpu_fp32'] = err
            if name == 'default_mode':
                chk['default_mode_selection'] = e.describe()
            chk[f'{name}_argmax_mismatch'] = int(mism.sum())
            # the tie band of the oracle (voxels whose own top-2 margin is within twice the measured error) next to the mismatch count
            # (VERDICT r4 item 1d): a mismatch inside it is a tie, one outside it would be a wrong class
            top2 = torch.topk(ref, 2, dim=1).values
            mg = (top2[:, 0] - top2[:, 1]).reshape(-1)
            chk[f'{name}_tie_band_voxels'] = int((mg <= 2 * err).sum())
            chk[f'{name}_argmax_mismatch_outside_tie_band'] = int((mism & (mg > 2 * err)).sum())
            if mism.any():
                chk[f'{name}_largest_oracle_margin_at_a_mismatch'] = float(mg[mism].max())
        chk['voxels'] = nvox
    return out, chk


def predict_2p5d_leg(dev, with_cpu):
    """The reference's own predict semantics (predict.py:79-112): one 128^3 uint8 block through the 2-D U-Net
    (4 levels, base 32, 1 -> 2 classes) slice by slice along the three axes -- strided views of the block, the head accumulating
    into [S, S, S, C] -- in the default prediction mode (fp16x2) and in fp16; plus the oracle's predict_block on the host cores."""
    import warnings
    from interactive_unet import predict as P
    from interactive_unet.unet import UNet
    S, C = 128, 2
    blk = synth_chunks(1, (S, S, S), 777, dev).reshape(S, S, S)
    out = torch.empty((S, S, S, C), dtype=torch.float32, device=dev)
    fpv2 = flops_per_voxel(2, 4, 32, 1, C)
    res = {'block': [S, S, S], 'axes': 3, 'semantics': 'predict.py:79-112 (2-D net over the slices along 3 axes, probabilities averaged)',
           'fwd_flop_per_block': 3 * fpv2 * S ** 3}
    probs = {}
    for name, kw in (('fp16x2', {}), ('fp16', {'act_dtype': 'fp16'})):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(num_classes=C, dim=2, pretrained=False, **kw)
        m.reset_parameters(seed=0)
        m = m.to(dev).eval()
        run = lambda: P.predict_block_device(m, blk, out, C, None, (0, 1, 2))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        # three groups of 10 blocks, every group reported, the fastest one quoted: a block is ~90 dependent launches of 10-100 us, and
        # a host thread that loses its core for a scheduler period (the box's CPU share is spent in bursts by the phases around this
        # leg: 16-thread torch CPU operators, numpy generators) stalls one group by tens of milliseconds -- seen as 7-10 ms per block
        # in one group where the others read 2.7
        reps, groups = 10, []
        for _ in range(3):
            t0 = time.time()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            groups.append((time.time() - t0) / reps * 1e3)
        ms = min(groups)
        res[name] = {'ms_per_block': round(ms, 3), 'ms_per_block_groups_of_10': [round(g, 3) for g in groups],
                     'voxels_per_s': round(S ** 3 / ms * 1e3, 1), 'tflops(algorithmic)': round(3 * fpv2 * S ** 3 / ms / 1e9, 1)}
        probs[name] = out.clone()
        if hasattr(m.engine('eval'), 'describe'):          # ('fp16x2' = the default split-precision mode: which form the calibration selected)
            res[name]['selected'] = m.engine('eval').describe()
        params = {k: t.detach().float().cpu() for k, t in m.named_tensors().items()}
        del m
    if with_cpu:
        from oracle import predict_ref, unet_ref
        cores = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(cores)
        fn = lambda b: unet_ref.forward(params, torch.from_numpy(b), dim=2).numpy()
        x = blk.cpu().numpy().astype(np.float32) / 255.0
        t0 = time.time()
        with torch.no_grad():
            want = predict_ref.predict_block(fn, x, C, 8, (0, 1, 2))
        dt = time.time() - t0
        res['cpu_baseline'] = {'value': round(S ** 3 / dt, 1), 'unit': 'voxels/s', 'cores': cores, 'kind': 'port',
                               'sample': f'the oracle predict_block (fp32 2-D U-Net, batch 8, 3 axes) on the same 128^3 block, torch CPU, {cores} threads'}
        for name in probs:
            d = (probs[name].cpu() - torch.from_numpy(want)).abs()
            res[name]['max_abs_prob_vs_cpu_fp32'] = float(d.max())
            res[name]['argmax_mismatch'] = int((probs[name].cpu().argmax(-1) != torch.from_numpy(want).argmax(-1)).sum())
    return res


def relaunch(args):
    """--gpus N without a launcher: start N ranks of this script under torch.distributed.run as CHILD processes
    (nothing in this process has touched the GPU) and hand back their exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--chunks', type=int, default=None, help='tiles per GPU per step (default: the workload\'s)')
    ap.add_argument('--c4-reps', type=int, default=1, help='c3 only: timed 1024^3 predictions appended to the line (0 = skip)')
    ap.add_argument('--c4-size', type=int, default=1024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--norm', default='batch', choices=['batch', 'group'], help="'group': GroupNorm(8) instead of BatchNorm after every stage conv "
                    "(north_star 'GroupNorm/BN'); statistics per (sample, group) at training and inference, nothing folds")
    ap.add_argument('--no-2p5d', action='store_true', help='skip legs.predict_2p5d (the reference-semantics 2.5-D block prediction)')
    ap.add_argument('--no-parity-mode', action='store_true', help='time the all-16-bit step only (no split-precision prediction leg)')
    ap.add_argument('--no-extras', action='store_true', help='c3 only: skip the compact c2 / c5 sub-objects of the line')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(relaunch(args))

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started {world} ranks')
    # rehearsal switches for a box with ONE GPU (the builder's): IUNET_BENCH_BACKEND=gloo IUNET_BENCH_SHARE_GPU=1 run the N ranks of
    # the whole distributed path on cuda:0 over gloo (which moves CUDA tensors through the host); never used by the driver
    backend = os.environ.get('IUNET_BENCH_BACKEND', 'nccl')
    if os.environ.get('IUNET_BENCH_SHARE_GPU'):
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a mismatched exchange schedule must end the run with a non-zero exit, not hang it: every collective / point-to-point wait of
        # the job fails after IUNET_BENCH_TIMEOUT_S seconds (default 300)
        tmo = datetime.timedelta(seconds=float(os.environ.get('IUNET_BENCH_TIMEOUT_S', '300')))
        if backend == 'nccl':
            os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
            dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    group = dist.group.WORLD if dist else None
    dev = torch.device('cuda', local)
    ranks_seen = 1
    if dist is not None:
        # proof that the communicator carries `world` ranks: an all-reduce of ones, reported as `rccl_ranks`
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        if ranks_seen != world:
            sys.exit(f'bench.py: the all-reduce of ones over the process group returned {ranks_seen}, expected {world}')

    out = run(args, args.workload, rank, world, dev, dist, group)
    if out is not None and rank == 0:
        out['rccl_ranks'] = ranks_seen if (dist is not None and backend == 'nccl') else None
        out['process_group'] = {'backend': backend if dist is not None else None, 'ranks_seen_by_all_reduce': ranks_seen}
        if world == 1 and args.workload == 'c3' and not args.no_extras:
            # the other single-GPU configurations of BASELINE.json in the SAME driver-run line, as compact sub-objects
            # (10 steps each; the full lines of those workloads are `--workload c2 / c5`)
            for wl in ('c2', 'c5'):
                sub = argparse.Namespace(**vars(args))
                sub.workload, sub.steps, sub.warmup, sub.chunks, sub.c4_reps, sub.no_2p5d = wl, 10, 3, None, 0, True
                try:
                    out[wl] = compact(run(sub, wl, rank, world, dev, dist, group))
                except Exception as e:                                         # the headline must not be lost to a side measurement
                    out[wl] = {'error': f'{type(e).__name__}: {e}'}
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def compact(o):
    """The sub-object of a side workload in the default line: step time, legs, roofline fraction, CPU baseline."""
    r = o['roofline']
    keep = {'workload': o['config']['workload'].split(';')[0], 'dtype': o['dtype'], 'headline_mode': o.get('headline_mode'), 'steps': o['steps'],
            'value': o['value'], 'unit': o['unit'], 'ms_per_step': o['ms_per_step'],
            'legs': {k: v for k, v in o['legs'].items() if k in ('train_ms', 'predict_ms', 'train_tflops_per_gpu(3x fwd)', 'predict_tflops_per_gpu')},
            'roofline': {k: r.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'ms_per_launch', 'traffic')},
            'cpu_baseline': o.get('cpu_baseline')}
    if 'in_situ' in r:
        keep['roofline']['in_situ_frac'] = r['in_situ'].get('frac')
    if o.get('throughput_mode'):
        keep['throughput_mode'] = {k: o['throughput_mode'].get(k) for k in ('dtype', 'value', 'ms_per_step', 'predict_ms', 'train_ms')}
    if 'parity' in o:
        keep['parity'] = {k: v for k, v in o['parity'].items() if k != 'vs_cpu_oracle'}
        if 'vs_cpu_oracle' in o['parity']:
            keep['parity']['vs_cpu_oracle'] = o['parity']['vs_cpu_oracle']
    return keep


def run(args, workload, rank, world, dev, dist, group):
    """One workload, timed by the bench contract; returns the JSON object on rank 0 (None elsewhere)."""
    from interactive_unet import _native as nv
    from interactive_unet import shard
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    import warnings

    cfg = dict(WORKLOADS[workload])
    if args.chunks is not None:
        cfg['chunks'] = args.chunks
    dim, levels, base, ncls, B = cfg['dim'], cfg['levels'], cfg['base'], cfg['ncls'], cfg['chunks']
    tile = tuple(cfg['tile'])
    S = tile[0]
    tvox = int(np.prod(tile))
    dtype = torch.bfloat16 if cfg['dtype'] == 'bf16' else torch.float16
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(lr=1e-4, num_classes=ncls, dim=dim, levels=levels, base=base, act_dtype=cfg['dtype'], pretrained=False,
                     weight_dtype=cfg['wq'], norm=args.norm)
    cfg['norm'] = args.norm
    model.reset_parameters(seed=0)                                      # random-init weights of the canonical architecture
    model = model.to(dev)
    fpv = flops_per_voxel(dim, levels, base, 1, ncls)
    legs = {'train': 0.0, 'predict': 0.0}
    leg_events = []
    info = {}
    ops = None

    # ------------------------------------------------------------------ the step of each workload
    if workload == 'c4':
        Vs = args.c4_size
        V = (Vs, Vs, Vs)
        bounds, _ = shard.slab_bounds(V[0], world)
        my_slab = synth_volume_slab(bounds[rank][0], bounds[rank][1], V[1], V[2], dev)
        ops = shard.NativeOps(model, ncls, S)

        def step(timed=False):
            t0 = time.time()
            _, st = shard.predict_volume_sharded(ops, my_slab, V, S, 0.25, group=group)
            info.update(st)
            if timed:
                torch.cuda.synchronize(); legs['predict'] += time.time() - t0
        unique_vox = float(np.prod(V))
        nblocks = len(shard.P.get_block_coordinates(np.array(V), S, 0.25)[0])
        processed_vox = float(nblocks) * tvox
        train_vox = 0.0
        scaling = 'strong'
    else:
        trainer = TrainEngine(model, lr=1e-4, loss_kind='mcc_ce', process_group=group)
        chunks = synth_chunks(B, tile, 1234 + rank, dev)
        X = chunks.reshape((B, 1) + tile)                                  # uint8, /255 in the first conv
        lab = (X.to(torch.int32) * ncls // 256).clamp_(max=ncls - 1)
        y = torch.cat([(lab == c) for c in range(ncls)], 1).to(torch.float16)       # loader contract: fp16 one-hot (loader.py:150-152)
        w = (torch.rand((B, 1) + tile, device=dev) > 0.1).to(torch.float16).expand((B, ncls) + tile).contiguous()
        y = y * w
        if dim == 3:
            # predict leg: ONE volume shared by all ranks, Z = 96*B*world + 32 planes of S x S -> exactly B overlapping
            # S^3 blocks per rank (predict.py:362-411 grid, overlap 0.25); every rank owns a z-slab, the slabs are
            # all-gathered and the pieces of the block probabilities exchanged over RCCL (shard.py)
            stride = int(S * 0.75)
            V = (stride * B * world + (S - stride), S, S)
            bounds, _ = shard.slab_bounds(V[0], world)
            my_slab = synth_chunks(1, tile, 4321 + rank, dev).reshape(tile)
            hh = bounds[rank][1] - bounds[rank][0]
            my_slab = my_slab.repeat(-(-hh // S), 1, 1)[:hh].contiguous()
            ops = shard.NativeOps(model, ncls, S)
            unique_vox = float(np.prod(V))
            processed_vox = float(B * world) * tvox

            def predict_leg():
                _, st = shard.predict_volume_sharded(ops, my_slab, V, S, 0.25, group=group)
                info.update(st)
        else:
            # 2-D: the slices of the batch through the forward + softmax + argmax (predict.py:16-47 per slice)
            V = (B * world,) + tile
            probs = torch.empty((B, ncls) + tile, dtype=torch.float32, device=dev)
            cls = torch.empty((B, tvox), dtype=torch.uint8, device=dev)
            unique_vox = processed_vox = float(B * world) * tvox

            def predict_leg():
                model.engine('eval').infer(chunks, (tvox, tvox, tvox, tile[1], 1), B, 1, tile[0], tile[1], probs=probs, cls=cls)
        train_vox = float(B * world) * tvox
        scaling = 'weak'

        def step(timed=False):
            # timed: the two legs by HIP events on the launch stream, no sync between them -- a sync would add the host's wake-up and
            # enqueue latency (~0.3 ms) to the leg behind it, which the K timed steps (the host runs ahead of the GPU) never pay
            if timed:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            trainer.train_step(X, y, w, sync=False)
            if timed:
                ev[1].record()
            model.engine('eval')                                            # re-packs the updated weights (BN folded)
            predict_leg()
            if timed:
                ev[2].record()
                leg_events.append(ev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def set_predict_mode(mode):
        """Prediction engine of the following steps: 'fp16x2' (split precision, the tolerance-meeting mode) or the workload's 16-bit dtype."""
        nonlocal ops
        model.infer_dtype, model._engines, model._packed_sig = mode, {}, None
        if ops is not None:
            ops = shard.NativeOps(model, ncls, S)                 # (step / predict_leg look `ops` up at call time)

    def timed_region(c4_reps):
        """warm-up, barrier, K steps, barrier, max over ranks; then the leg split in separate steps (the timed region has no extra
        syncs) and, for c3, the C4 volume."""
        nonlocal legs
        legs = {'train': 0.0, 'predict': 0.0}
        for _ in range(max(1, args.warmup)):
            step()
        barrier()
        t0 = time.time()
        for _ in range(args.steps):
            step()
        barrier()
        dt = max_over_ranks(time.time() - t0)
        nleg = 4 if workload != 'c4' else 1          # (2 steps read the prediction leg anywhere in 2.08-2.43 ms on one box)
        leg_events.clear()
        for _ in range(nleg):
            step(timed=True)
        torch.cuda.synchronize()
        for ev in leg_events:
            legs['train'] += ev[0].elapsed_time(ev[1]) * 1e-3
            legs['predict'] += ev[1].elapsed_time(ev[2]) * 1e-3
        res = {'dt': dt, 'train': legs['train'] / nleg, 'predict': legs['predict'] / nleg, 'c4': None}
        if workload == 'c3' and c4_reps > 0:
            Vs = args.c4_size
            V4 = (Vs, Vs, Vs)
            b4, _ = shard.slab_bounds(Vs, world)
            slab4 = synth_volume_slab(b4[rank][0], b4[rank][1], Vs, Vs, dev)
            st4 = {}
            run4 = lambda: st4.update(shard.predict_volume_sharded(ops, slab4, V4, S, 0.25, group=group)[1])
            run4()                                                             # warm-up (allocations)
            barrier()
            t4 = time.time()
            for _ in range(c4_reps):
                run4()
            barrier()
            d4 = max_over_ranks(time.time() - t4) / c4_reps
            nb4 = len(shard.P.get_block_coordinates(np.array(V4), S, 0.25)[0])
            res['c4'] = {'volume': list(V4), 'blocks': nb4, 'reps': c4_reps, 'seconds_per_volume': round(d4, 4), 'scaling': 'strong',
                         'volume_voxels_per_s': round(Vs ** 3 / d4, 1), 'processed_voxels_per_s': round(nb4 * tvox / d4, 1),
                         'tflops': round(fpv * nb4 * tvox / d4 / 1e12, 1), 'blocks_rank0': st4.get('blocks'),
                         'exchange_bytes_sent_rank0': st4.get('bytes_sent'), 'pieces_blended_rank0': st4.get('pieces_blended')}
            del slab4
        return res

    def probe_layer(n_steps, with_train):
        """HIP events around the dec0.conv1 launches of `n_steps` real steps (prediction engine; with_train: the training forward too)."""
        events = []
        eng = model.engine('eval')
        eng.probe = {'name': 'dec0.conv1', 'events': events}
        if with_train:
            trainer.probe = eng.probe
        for _ in range(n_steps):
            step()
            if len(events) > 400:
                break
        torch.cuda.synchronize()
        eng.probe = None
        if with_train:
            trainer.probe = None
        return events

    # ------------------------------------------------------------------ the headline region.  For the BatchNorm networks without
    # quantised weights it is the COMPLIANT pairing: training in the workload's 16-bit dtype (the reference trains
    # under '16-mixed', trainer.py:59) + prediction in the tolerance-meeting split-precision mode (fp16x2: logits within 1e-3 of the CPU
    # fp32 path, what UNet() predicts in by default; the reference predicts in fp32, predict.py:30-35).  The all-16-bit step is timed
    # afterwards, exactly the same way, as `throughput_mode`.
    compliant = not cfg['wq'] and not args.no_parity_mode
    if compliant:
        set_predict_mode('fp16x2')
    head = timed_region(args.c4_reps)
    selection = model.engine('eval').describe() if compliant and hasattr(model.engine('eval'), 'describe') else None
    dt = head['dt']
    value = (train_vox + unique_vox) * args.steps / dt
    c4 = head['c4']
    predict_events = probe_layer(1 if workload == 'c4' else 6, False) if compliant else []

    # ------------------------------------------------------------------ throughput mode (prediction in the 16-bit dtype) + the roofline
    # layer as the 16-bit step launches it
    tm = None
    if compliant:
        set_predict_mode(dtype)
        tm = timed_region(min(1, args.c4_reps))
    probe_train = workload != 'c4' and not cfg['wq']      # C5 trains on the 16-bit kernel: only its fp8 prediction launches count
    probe_events = probe_layer(1 if workload == 'c4' else 10, probe_train)

    out = None
    if rank == 0:
        # `roofline` top level = the layer alone, back to back (the figure `rocprofv3 --kernel-trace --stats` of tools/bench_conv.py
        # reproduces: profiles/r03_roofline_*); the launches inside real steps are the `in_situ` sub-object
        if probe_events:
            in_situ = conv_roofline_in_situ(nv, cfg, workload, dtype, probe_events[:400])
            roof = conv_roofline_back_to_back(nv, cfg, workload, dtype, in_situ['tiles_per_launch'])
            roof['traffic'] = in_situ.pop('traffic')
            roof['in_situ'] = in_situ
        else:                                     # (GroupNorm: the engines' timing hook sits in the BatchNorm launch path)
            roof = conv_roofline_back_to_back(nv, cfg, workload, dtype, max(1, B if dim == 3 else B))
            roof['traffic'] = pmc_traffic(workload, roof['tiles_per_launch'])
        if dim == 3 and roof['tiles_per_launch'] != 1:
            roof['back_to_back_1_tile'] = conv_roofline_back_to_back(nv, cfg, workload, dtype, 1)
        tile_s = 'x'.join(str(s) for s in tile)
        if workload == 'c4':
            desc = (f'C4: tiled prediction of one {V[0]}^3 uint8 volume (generated on the device from the voxel coordinates), '
                    f'{nblocks} blocks of {S}^3, overlap 0.25, 3-D U-Net {levels}-level base {base}, 1->{ncls} classes, {cfg["dtype"]}; '
                    f'one step = one whole-volume prediction sharded over the {world} rank(s): slab all-gather, block forwards + '
                    f'softmax, point-to-point exchange of the probability pieces in 8 rounds overlapped with the forwards, blend in '
                    f'flat block order by the slab owners, normalise/quantise; wall time includes every exchange')
        else:
            desc = (f'{workload.upper()}: {dim}-D U-Net {levels}-level base {base}{", GroupNorm(8)" if args.norm == "group" else ""}, 1->{ncls} classes, {cfg["dtype"]}'
                    f'{", inference weights e4m3" if cfg["wq"] else ""}; {B} x {tile_s} uint8 tiles per GPU per step; step = 1 training '
                    f'step (forward with BatchNorm batch stats, MCC+CE loss, backward, AdamW, weight re-pack; gradients all-reduced '
                    f'over RCCL for N > 1) on the {B} tiles + ' +
                    (f'tiled prediction of one {V[0]}x{S}x{S} uint8 volume shared by all ranks = {B} overlapping {S}^3 blocks per GPU '
                     f'(reflect-padded block gather, forward + softmax, Gaussian blend, piece exchange over RCCL for N > 1, '
                     f'normalise/quantise)' if dim == 3 else f'forward + softmax + argmax of the same {B} slices') +
                    '; value = (training tile voxels + UNIQUE predicted voxels) / s')
        out = {
            'metric': 'voxels/sec (train step + full-volume predict) on 128^3 chunks',
            'value': round(value, 1), 'unit': 'voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': cfg['dtype'], 'data': 'synthetic',
            'config': {'workload': desc, 'predict_volume': list(V), 'predict_blocks_per_gpu': info.get('blocks'),
                       'predict_exchange_bytes_sent_rank0': info.get('bytes_sent'),
                       'tiles_per_gpu': B, 'tile': list(tile), 'levels': levels, 'base': base,
                       'train_voxels_per_step': train_vox, 'predict_unique_voxels_per_step': unique_vox,
                       'predict_processed_voxels_per_step': processed_vox, 'fwd_flop_per_voxel': fpv},
            'roofline': roof,
        }
        def leg_fields(r):
            lg = {}
            if r['train'] > 0:
                tt = r['train']
                lg.update({'train_ms': round(tt * 1e3, 3), 'train_voxels_per_s_per_gpu': round(train_vox / world / tt, 1),
                           'train_tflops_per_gpu(3x fwd)': round(3 * fpv * train_vox / world / tt / 1e12, 1)})
            tp = r['predict']
            lg.update({'predict_ms': round(tp * 1e3, 3), 'predict_volume_voxels_per_s': round(unique_vox / tp, 1),
                       'predict_processed_voxels_per_s': round(processed_vox / tp, 1),
                       'predict_tflops_per_gpu': round(fpv * processed_vox / world / tp / 1e12, 1)})
            return lg
        out['legs'] = leg_fields(head)
        if c4 is not None:
            out['c4'] = c4
        x2m_on = compliant and selection is not None and selection['form'] == 'x2m'      # the form engine_auto.EngineAuto selected
        x2_name = ('fp16x2 with the cross terms on the fp8 matrix cores (x2m: x_hi w_hi on v_mfma_f32_16x16x32_f16 + [x_lo8 | x_hi8][w_hi8 | w_lo8] on '
                   'v_mfma_f32_16x16x128_f8f6f4, fp32 accumulate)') if x2m_on else 'fp16x2 (split precision: fp16 hi + lo words, fp32 accumulate)'
        if compliant:
            out['headline_mode'] = (f'compliant pairing: training {cfg["dtype"]} (trainer.py:59 trains under 16-mixed) + prediction {x2_name}: logits within '
                                    f'1e-3 of the CPU fp32 path (predict.py:30-35 predicts in fp32); the all-16-bit step is `throughput_mode`')
            # the form the default engine SELECTED for this model (x2m against fp16x2 on a calibration tile; engine_auto.py) and the figure behind it
            out['config']['predict_dtype'] = f"fp16x2 ({selection['form']})" if selection else 'fp16x2'
            out['config']['predict_mode_selection'] = selection
            if predict_events:
                ps = conv_roofline_in_situ(nv, cfg, workload, torch.float16, predict_events[:400])
                ps.pop('traffic', None)
                if dim == 3:
                    ps['traffic'] = pmc_traffic('x2m' if x2m_on else 'x2', ps['tiles_per_launch'])
                    bpe = 1.5 if x2m_on else 2.0                      # bytes per element against the 16-bit kernel's 2: x2m 3 (hi word + lo8 byte), fp16x2 4 (hi + lo words)
                    ps['algorithmic_bytes_per_launch'] = int(ps['algorithmic_bytes_per_launch'] * bpe)
                    ps['hbm_gbs_algorithmic'] = round(bpe * ps['hbm_gbs_algorithmic'], 1)
                ps['kernel'] = (('conv3_x2m_kernel' if dim == 3 else 'conv2_x2m_kernel') if x2m_on else 'split-precision conv (conv3_v4_kernel<f16,...,SPL>)') + ', ' + ps['kernel'].split('(', 1)[1].rstrip(')')
                # matrix work per multiply-add: fp16x2 three 16-bit products; x2m one 16-bit product + two fp8 products at twice the rate =
                # the time of two 16-bit products at peak
                units = 2 if x2m_on else 3
                ps['matrix_step_units_per_mac'] = units
                ps['mfma_work_tflops(16-bit equivalent)'] = round(units * ps['achieved'], 1) if ps['unit'] == 'TFLOP/s' else round(units * ps.get('tflops', 0.0), 1)
                ps['mfma_work_frac'] = round(ps['mfma_work_tflops(16-bit equivalent)'] / MFMA_PEAK_TFLOPS, 4)
                roof['predict_kernel'] = ps
            t = leg_fields(tm)
            out['throughput_mode'] = {'dtype': f'train {cfg["dtype"]} + predict {cfg["dtype"]}' if train_vox else f'predict {cfg["dtype"]}',
                                      'value': round((train_vox + unique_vox) * args.steps / tm['dt'], 1), 'unit': 'voxels/s', 'steps': args.steps,
                                      'ms_per_step': round(tm['dt'] / args.steps * 1e3, 3), **t,
                                      'deviation': 'see `parity`: the 16-bit prediction does not meet 1e-3 on the logits'}
            if tm['c4'] is not None:
                out['throughput_mode']['c4'] = tm['c4']
        if workload == 'c4':
            probe = my_slab[:S, :S, :S].contiguous()
        else:
            probe = chunks[0]
        out['parity'], _ = native_parity(model, cfg, probe)
        # (the 2.5-D leg's device timings come BEFORE the host baseline: its ~90 launches per block are sensitive to the host cores, and the
        #  baseline's 16 torch threads keep spinning for a while after their last operator -- one run read 6.8 ms instead of 3.0 that way)
        if workload in ('c3', 'c2') and not args.no_2p5d:
            out['legs']['predict_2p5d'] = predict_2p5d_leg(dev, world == 1 and not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'], chk = cpu_baseline(cfg, model, probe)
            if chk:
                out['parity']['vs_cpu_oracle'] = chk
    return out


if __name__ == '__main__':
    main()
