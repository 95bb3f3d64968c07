"""How far is each prediction mode from the CPU fp32 oracle on a TRAINED network?

    python tools/trained_parity.py [--dim 2|3] [--steps N] [--lr LR] [--target 20]

Trains the canonical net natively (fp16 TrainEngine, MCC+CE) on separable synthetic labels (label = smooth image > threshold) until
max |logit| of the oracle-sized evaluation tile reaches --target, then prints, at several points of the trajectory, the logit scale and
max |logit - fp32 oracle| of x2m, fp16x2 and the fp32 mode on a 512^2 slice (2-D) / a 64^3 or 128^3 chunk (3-D), plus the on-device
x2m-vs-fp16x2 difference the selection rule of engine_auto.py measures.
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))


def smooth(shape, seed, sigma=6):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    v = ndimage.gaussian_filter(rng.random(shape), sigma)
    v = (v - v.min()) / (v.max() - v.min())
    return (v * 254 + 1).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dim', type=int, default=2)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--lr', type=float, default=2e-3)
    ap.add_argument('--target', type=float, default=20.0)
    ap.add_argument('--eval-every', type=int, default=50)
    ap.add_argument('--eval-size', type=int, default=None)
    ap.add_argument('--loss', default='mcc_ce')
    ap.add_argument('--wd', type=float, default=1e-2)
    args = ap.parse_args()
    from interactive_unet.unet import UNet
    from interactive_unet.train_engine import TrainEngine
    from interactive_unet.engine_x2 import EngineX2
    from interactive_unet.engine_f32 import EngineF32
    from oracle import unet_ref
    dim = args.dim
    dev = torch.device('cuda')
    tshape = (256, 256) if dim == 2 else (64, 64, 64)
    B = 8 if dim == 2 else 2
    es = args.eval_size or (512 if dim == 2 else 64)
    eshape = (es,) * dim
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(lr=args.lr, dim=dim, act_dtype='fp16', pretrained=False)
    model.reset_parameters(seed=0)
    model = model.to(dev)
    te = TrainEngine(model, lr=args.lr, loss_kind=args.loss, weight_decay=args.wd)
    imgs = np.stack([smooth(tshape, 100 + i) for i in range(4 * B)])[:, None]
    X = torch.tensor(imgs).to(dev)
    lab = (X > 127)
    Y = torch.cat([~lab, lab], 1).to(torch.float16)
    Wt = torch.ones_like(Y)
    ev = torch.tensor(smooth(eshape, 999))[None, None]
    D, H, W = eshape if dim == 3 else (1,) + eshape
    vox = D * H * W

    def evaluate(step):
        params = {k: t.detach().float().cpu() for k, t in model.named_tensors().items()}
        t0 = time.time()
        with torch.no_grad():
            ref = unet_ref.forward_logits(params, ev.float() / 255.0, dim=dim)
        tcpu = time.time() - t0
        top2 = torch.topk(ref, 2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1]).reshape(-1)
        want = ref.argmax(1).reshape(-1)
        row = {}
        lgs = {}
        for nm in ('x2m', 'fp16x2', 'fp32'):
            if nm == 'fp32':
                e = EngineF32(dim, 4, 32, 1, 2, dev)
            else:
                e = EngineX2(dim, 4, 32, 1, 2, dev, mixed=(nm == 'x2m'))
            e.load_eval(model.named_tensors())
            lg = torch.empty((1, 2) + eshape, device=dev)
            cl = torch.empty((1, vox), dtype=torch.uint8, device=dev)
            e.infer(ev.to(dev), (vox, vox, H * W, W, 1), 1, D, H, W, logits=lg, cls=cl)
            torch.cuda.synchronize()
            err = (lg.cpu() - ref).abs().max().item()
            mism = cl.cpu().long().reshape(-1) != want
            ties = int((margin <= 2 * err + 1e-7).sum())
            sat = e.saturated() if hasattr(e, 'saturated') else False
            row[nm] = (err, int(mism.sum()), ties, float(margin[mism].max()) if mism.any() else 0.0, sat)
            lgs[nm] = lg
            del e
        scale = ref.abs().max().item()
        dd = (lgs['x2m'] - lgs['fp16x2']).abs().max().item()
        print(f'[step {step:4d}] logit scale {scale:7.2f} (cpu {tcpu:.1f}s) | ' + ' | '.join(
            f'{nm}: err {r[0]:.2e} rel {r[0] / scale:.1e} mism {r[1]}/{r[2]} ties maxmargin {r[3]:.1e}{" SAT" if r[4] else ""}' for nm, r in row.items())
            + f' | x2m-vs-fp16x2 on device {dd:.2e}', flush=True)
        return scale

    evaluate(0)
    for s in range(1, args.steps + 1):
        i = (s % 4) * B
        o = te.train_step(X[i:i + B], Y[i:i + B], Wt[i:i + B], sync=(s % args.eval_every == 0))
        if s % args.eval_every == 0:
            print(f'    loss {o["Loss"]:.4f} dice {o["Dice"]:.4f} mcc {o["MCC"]:.4f}', flush=True)
            if evaluate(s) >= args.target and s >= 2 * args.eval_every:
                break


if __name__ == '__main__':
    main()
