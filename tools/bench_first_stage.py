"""First encoder stage of the 2-D x2m forward: iunet_x2m_first_stage_fwd (one launch, the first conv computed by the second conv's loader
waves) against iunet_x2m_first_conv_fwd + iunet_x2m_conv_fwd.   python tools/bench_first_stage.py [N] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv
from tests.test_gpu_x2 import _prep_conv
from tests.test_gpu_x2m import _prep, A

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
c, vox = 32, S * S
g = torch.Generator().manual_seed(0)
x = torch.randint(0, 256, (N, S, S), generator=g, dtype=torch.uint8).cuda()
st = nv.ll_array((vox, vox, vox, S, 1))
fw, fosc, fb = _prep_conv(nv, torch.randn((c, 1, 3, 3), generator=g) * 0.4)
w16, w8, osc, bias, _ = _prep(nv, torch.randn((c, c, 3, 3), generator=g) * 0.08)
y = torch.zeros(N * c * vox, dtype=torch.float16, device='cuda'); y8 = torch.zeros(N * 2 * c * vox, dtype=torch.uint8, device='cuda')
a = torch.zeros_like(y); a8 = torch.zeros_like(y8)
s = nv.stream()


def timeit(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


fused = lambda: nv.call('iunet_x2m_first_stage_fwd', nv.ptr(x), 2, st, nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb), A, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox,
                        None, 0, None, 0, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, S, S, None, s)
first = lambda: nv.call('iunet_x2m_first_conv_fwd', 2, nv.ptr(x), 2, st, nv.ptr(a), c * vox, -1, nv.ptr(a8), 2 * c * vox, nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb), A,
                        N, 1, S, S, 1, c, 1, None, s)
conv = lambda: nv.call('iunet_x2m_conv_fwd', 2, nv.ptr(a), c * vox, nv.ptr(a8), 2 * c * vox, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox,
                       nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, 1, S, S, c, c, 2, None, s)
tf, t1, t2 = timeit(fused), timeit(first), timeit(conv)
print(f'{N} x {S}^2: first stage in one launch {tf:.1f} us against {t1:.1f} (first conv) + {t2:.1f} (second conv) = {t1 + t2:.1f} us')
# ... and with the stage's max-pool: what the network's encoder stage 0 runs
pv = vox // 4
p = torch.zeros(N * c * pv, dtype=torch.float16, device='cuda'); p8 = torch.zeros(N * 2 * c * pv, dtype=torch.uint8, device='cuda')
fused_pool = lambda: nv.call('iunet_x2m_first_stage_fwd', nv.ptr(x), 2, st, nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb), A, nv.ptr(y), c * vox, -1, nv.ptr(y8),
                             2 * c * vox, nv.ptr(p), c * pv, nv.ptr(p8), 2 * c * pv, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, S, S, None, s)
conv_pool = lambda: nv.call('iunet_x2m_conv_pool_fwd', 2, nv.ptr(a), c * vox, nv.ptr(a8), 2 * c * vox, nv.ptr(y), c * vox, -1, nv.ptr(y8), 2 * c * vox,
                            nv.ptr(p), c * pv, nv.ptr(p8), 2 * c * pv, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), N, 1, S, S, c, c, 2, None, s)
pool = lambda: nv.call('iunet_x2m_maxpool_fwd', 2, nv.ptr(y), c * vox, nv.ptr(y8), 2 * c * vox, nv.ptr(p), c * pv, nv.ptr(p8), 2 * c * pv, c, N, 1, S // 2,
                       S // 2, s)
tfp, tcp, tp = timeit(fused_pool), timeit(conv_pool), timeit(pool)
print(f'{N} x {S}^2 with the pool: all in one launch {tfp:.1f} us; first stage + pool launch {tf:.1f} + {tp:.1f} = {tf + tp:.1f} us; '
      f'first conv + pooled conv {t1:.1f} + {tcp:.1f} = {t1 + tcp:.1f} us; three launches {t1 + t2 + tp:.1f} us')
