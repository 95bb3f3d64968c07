"""Interactive latency of one slice (predict.py:16-47's operating point: one 128..512-pixel slice per call): the Python-sequenced
engine (one ctypes call per kernel), the C++-sequenced handle (iunet_net_forward: one call), and a captured HIP graph of either."""
import ctypes, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv
from interactive_unet.unet import UNet

def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

for mode in ('fp16', 'fp16x2'):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=2, dim=2, act_dtype=mode, pretrained=False).cuda().eval()
    eng = m.engine('eval')
    for S in (128, 256, 512):
        x = torch.randint(1, 255, (S, S), dtype=torch.uint8, device='cuda')
        probs = torch.empty(1, 2, S, S, device='cuda'); cls = torch.empty(1, S * S, dtype=torch.uint8, device='cuda')
        run = lambda: eng.infer(x, (S * S, S * S, S * S, S, 1), 1, 1, S, S, probs=probs, cls=cls)
        py = t(run)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run()
        gr = t(g.replay)
        # C++-sequenced handle
        h = ctypes.c_void_p()
        nv.call('iunet_net_create', 2, 4, 32, 1, 2, (3 if getattr(m.engine('eval'), 'mixed', False) else 2) if mode == 'fp16x2' else 0, 0.0, ctypes.byref(h))      # (fp16x2: the engine's own form -- mode 3, cross terms on the fp8 matrix cores, by default)
        flat = torch.empty(nv.lib().iunet_net_num_params(h), device='cuda')
        off = 0
        for name, tns in m.named_tensors().items():
            flat[off:off + tns.numel()] = tns.detach().reshape(-1); off += tns.numel()
        packed = torch.empty(nv.lib().iunet_net_packed_bytes(h), dtype=torch.uint8, device='cuda')
        nv.call('iunet_net_load', h, nv.ptr(flat), nv.ptr(packed), nv.stream())
        ws = torch.empty(nv.lib().iunet_net_workspace_bytes(h, 1, 1, S, S), dtype=torch.uint8, device='cuda')
        st, os_ = nv.ll_array((S * S, S * S, S * S, S, 1)), nv.ll_array((2 * S * S, S * S, S * S, S, 1))
        cc = lambda: nv.call('iunet_net_forward', h, nv.ptr(x), 2, st, 1, 1, S, S, nv.ptr(ws), None, nv.ptr(probs), nv.ptr(cls), os_, 1.0, 0, nv.stream())
        cpp = t(cc)
        print(f'{mode:7s} {S}^2: python-sequenced {py:.3f} ms, C++-sequenced {cpp:.3f} ms, HIP graph replay {gr:.3f} ms', flush=True)
        nv.lib().iunet_net_destroy(h)
