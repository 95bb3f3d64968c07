"""The head's backward + the last conv's BatchNorm backward at the C3 step's size (2 x 128^3, 32 channels, 2 classes, bf16): the two-pass
fused form (iunet_head_bn_bwd) against the sequence it replaces (iunet_head_loss_bwd_act -> iunet_bn_relu_bwd).   python tools/bench_head_bn.py [N] [vox]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import _native as nv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vox = int(sys.argv[2]) if len(sys.argv) > 2 else 128 ** 3
T, C0, ncls = torch.bfloat16, 32, 2
dt = nv.DTYPE_CODE[T]
g = torch.Generator(device='cuda').manual_seed(0)
yb = (torch.randn(N * C0 * vox, device='cuda', generator=g)).to(T)
w = torch.randn(ncls, C0, device='cuda', generator=g) * 0.3
b = torch.zeros(ncls, device='cuda')
tgt = (torch.rand((N, ncls, vox), device='cuda', generator=g) > 0.5).to(torch.float16)
wt = torch.ones((N, ncls, vox), device='cuda', dtype=torch.float16)
gamma = torch.ones(C0, device='cuda'); mean = torch.zeros(C0, device='cuda'); invstd = torch.ones(C0, device='cuda')
scale = gamma * invstd; shift = torch.zeros(C0, device='cuda')
coef = torch.tensor([[1e-7, 2e-7, 1e-7], [1e-7, -2e-7, 1e-7]], device='cuda')
parts = nv.lib().iunet_head_loss_bwd_num_parts(N, vox, ncls, C0)
dy = torch.empty(N * C0 * vox, dtype=T, device='cuda'); dz = torch.empty_like(dy)
hslab = torch.empty(parts * ncls * (C0 + 1), device='cuda'); bnslab = torch.empty(parts * C0 * 2, device='cuda'); bncoef = torch.empty(3 * C0, device='cuda')
dgam, dbet = torch.empty(C0, device='cuda'), torch.empty(C0, device='cuda')
dlbuf = torch.empty(N * vox * ncls, device='cuda')
s = nv.stream()
def fused():
    nv.call('iunet_head_bn_bwd', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), 1024.0, None,
            nv.ptr(scale), nv.ptr(shift), nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(dgam), nv.ptr(dbet), nv.ptr(dy), C0 * vox,
            nv.ptr(hslab), nv.ptr(bnslab), nv.ptr(bncoef), nv.ptr(dlbuf), N, vox, s)
def seq():
    nv.call('iunet_head_loss_bwd_act', dt, nv.ptr(yb), C0 * vox, C0, nv.ptr(w), nv.ptr(b), ncls, nv.ptr(tgt), nv.ptr(wt), 1, nv.ptr(coef), 1024.0,
            nv.ptr(dz), C0 * vox, nv.ptr(hslab), nv.ptr(scale), nv.ptr(shift), N, vox, s)
    nv.call('iunet_bn_relu_bwd', dt, nv.ptr(dz), C0 * vox, None, 0, nv.ptr(yb), C0 * vox, nv.ptr(dy), C0 * vox, nv.ptr(mean), nv.ptr(invstd),
            nv.ptr(gamma), nv.ptr(scale), nv.ptr(shift), nv.ptr(dgam), nv.ptr(dbet), nv.ptr(bnslab), nv.ptr(bncoef), C0, N, vox, s)
for name, fn in (('two passes over y (iunet_head_bn_bwd)', fused), ('head_loss_bwd_act + bn_relu_bwd', seq)) * 2:
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  ({N} x {vox} voxels)')
