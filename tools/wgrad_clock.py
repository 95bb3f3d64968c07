"""In-kernel clock and cycles per tile of conv3_wgrad_v2_kernel (diagnostic build with -DIUNET_STAMPS only):
    IUNET_LIB=<stamped .so> python tools/wgrad_clock.py"""
import ctypes, sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'interactive-unet_amd'))
from interactive_unet import _native as nv
T=torch.bfloat16; dt=nv.DTYPE_CODE[T]; nd=3
for (cin,cout,S) in [(32,32,128),(64,32,128)]:
    vox=S**3; n=1
    x=(torch.randn(n*cin*vox,device='cuda')*0.5).to(T); dy=(torch.randn(n*cout*vox,device='cuda')*0.5).to(T)
    nfl=nv.lib().iunet_conv3_wgrad_slab_floats(nd,n,S,S,S,cin,cout)
    slab=torch.empty(nfl,device='cuda'); dW=torch.empty(cout*cin*27,device='cuda')
    g=lambda: nv.call('iunet_conv3_wgrad', dt, nd, nv.ptr(x), cin*vox, nv.ptr(dy), cout*vox, nv.ptr(slab), nv.ptr(dW), 1.0, n,S,S,S,cin,cout, nv.stream())
    for _ in range(30): g()
    torch.cuda.synchronize()
    out=(ctypes.c_ulonglong*2048)()
    nv.lib().iunet_wg2_stamps_read(out)
    a=np.array(out[:]).reshape(512,4)
    a=a[a[:,2]>0]
    clk=a[:,0]/a[:,1]*100.0  # MHz
    print(cin,cout,'WGs',len(a),'tiles',a[0,2],'clock MHz median',np.median(clk),'min',clk.min(),'max',clk.max(),'cycles/tile',np.median(a[:,0]/a[:,2]), 'us', np.median(a[:,1])/100)
