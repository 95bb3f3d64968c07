"""2-D training step (the reference UI's configuration: 512^2 slices, fp16, MCC+CE; app.py:203-210 offers batch 1 .. 32): wall time per
step, GPU time per step and the host time to enqueue one, sequenced from Python (train_engine.TrainEngine) and as ONE C call
(csrc/train_net.hip: iunet_train_step).   python tools/bench_train2d.py [batch ...]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import numpy as np, torch
from interactive_unet.unet import UNet
from interactive_unet.train_engine import TrainEngine
for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    for mode in ('python', 'c'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = UNet(num_classes=2, dim=2, act_dtype='fp16', pretrained=False).cuda()
        te = TrainEngine(m, lr=1e-4, loss_kind='mcc_ce')
        te.use_handle = mode == 'c'
        g = torch.Generator(device='cuda').manual_seed(0)
        X = torch.randint(1, 255, (B, 1, 512, 512), dtype=torch.uint8, device='cuda', generator=g)
        lab = X > 127
        y = torch.cat([~lab, lab], 1).half()
        w = torch.ones_like(y)
        for _ in range(4): te.train_step(X, y, w, sync=False)
        torch.cuda.synchronize(); t0 = time.time()
        n = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): te.train_step(X, y, w, sync=False)
        e1.record(); torch.cuda.synchronize()
        wall = (time.time() - t0) / n * 1e3
        gpu = e0.elapsed_time(e1) / n
        t0 = time.time()
        for _ in range(n): te.train_step(X, y, w, sync=False)
        host = (time.time() - t0) / n * 1e3
        torch.cuda.synchronize()
        print(f'2-D train step, batch {B} x 512^2 fp16, sequenced from {mode:6s}: wall {wall:.3f} ms/step, GPU (events) {gpu:.3f} ms/step, host time to '
              f'enqueue one step {host:.3f} ms, {B * 512 * 512 / wall / 1e3:.1f} Mvox/s, {3 * 280256 * B * 512 * 512 / wall / 1e9:.0f} TFLOP/s (3x fwd)', flush=True)
        del te, m
