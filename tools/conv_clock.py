"""In-kernel clock and cycles per step of conv3_v4_kernel (diagnostic build with -DIUNET_STAMPS only):
    IUNET_LIB=<stamped .so> python tools/conv_clock.py [cin:cout:size ...]"""
import ctypes, sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'interactive-unet_amd'))
from interactive_unet import _native as nv
T = torch.bfloat16; dt = nv.DTYPE_CODE[T]; nd = 3
shapes = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or [(32, 32, 128), (64, 32, 128), (128, 64, 64)]
for cin, cout, S in shapes:
    vox = S ** 3; n = 1
    x = (torch.randn(n * cin * vox, device='cuda') * 0.5).to(T)
    y = torch.empty(n * cout * vox, dtype=T, device='cuda')
    w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
    wpk = torch.empty(nv.pack_conv3_elems(cout, cin, 27, 2), dtype=T, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    nv.call('iunet_pack_conv3', dt, nv.ptr(w), None, nv.ptr(wpk), cout, cin, 27, 2, nv.stream())
    f = lambda: nv.call('iunet_conv3_fwd', dt, nd, nv.ptr(x), cin * vox, nv.ptr(y), cout * vox, nv.ptr(wpk), nv.ptr(bias), None,
                        n, S, S, S, cin, cout, 2, 2, nv.stream())
    for _ in range(30): f()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 2048)()
    nv.lib().iunet_v4_stamps_read(out)
    a = np.array(out[:]).reshape(512, 4); a = a[a[:, 2] > 0]
    clk = a[:, 0] / a[:, 1] * 100.0
    print(f'{cin}->{cout} @{S}^3: WGs {len(a)} steps {int(a[0, 2])} clock MHz median {np.median(clk):.0f} (min {clk.min():.0f} max {clk.max():.0f}) '
          f'cycles/step {np.median(a[:, 0] / a[:, 2]):.0f} loop us {np.median(a[:, 1]) / 100:.1f}')
