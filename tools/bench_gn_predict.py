"""One 128^3 prediction forward of the canonical 3-D net with BatchNorm (folded) and with GroupNorm(8) (raw conv output + three normalisation passes per
conv), in both split-precision forms.   python tools/bench_gn_predict.py"""
import sys, time, torch
sys.path.insert(0,'interactive-unet_amd'); sys.path.insert(0,'.')
import warnings
from interactive_unet.unet import UNet
from interactive_unet.engine_x2 import EngineX2
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m=UNet(dim=3, pretrained=False)
m.reset_parameters(seed=1)
p={k:v.cuda() for k,v in m.named_tensors().items()}
x=torch.randint(1,255,(1,1,128,128,128),dtype=torch.uint8,device='cuda')
v=128**3
for norm in ('batch','group'):
    for mixed in (True, False):
        e=EngineX2(dim=3, norm=norm, mixed=mixed); e.load_eval(p)
        pr=torch.empty((1,2,128,128,128),device='cuda')
        for _ in range(3): e.infer(x,(v,v,128*128,128,1),1,128,128,128,probs=pr)
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(10): e.infer(x,(v,v,128*128,128,1),1,128,128,128,probs=pr)
        torch.cuda.synchronize(); print(f'128^3 forward, norm {norm}, {"x2m" if mixed else "fp16x2"}: {(time.time()-t0)/10*1e3:.3f} ms', flush=True)
