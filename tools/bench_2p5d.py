"""2.5-D block prediction (predict.py:79-112): one 128^3 block through the 2-D net along 3 axes -- the three axes as ONE batch of views
against three forwards (IUNET_2P5D_SEQUENTIAL=1).   python tools/bench_2p5d.py [S]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import predict as P
from interactive_unet.unet import UNet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device='cuda').manual_seed(0)
blk = torch.randint(1, 255, (S, S, S), dtype=torch.uint8, device='cuda', generator=g)
out = torch.empty((S, S, S, 2), device='cuda')
for kw in ({}, {'act_dtype': 'fp16'}):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=2, dim=2, pretrained=False, **kw)
    m.reset_parameters(seed=0)
    m = m.cuda().eval()
    for seq in ('', '1'):
        if seq:
            os.environ['IUNET_2P5D_SEQUENTIAL'] = '1'
        else:
            os.environ.pop('IUNET_2P5D_SEQUENTIAL', None)
        for _ in range(3):
            P.predict_block_device(m, blk, out, 2, None, (0, 1, 2))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            for _ in range(10):
                P.predict_block_device(m, blk, out, 2, None, (0, 1, 2))
            torch.cuda.synchronize()
            best = min(best, (time.time() - t0) / 10 * 1e3)
        eng = m.engine('eval')
        print(f'{S}^3 block, {getattr(eng, "form", kw.get("act_dtype"))}, {"three forwards" if seq else "one batch of views"}: {best:.3f} ms per block', flush=True)
