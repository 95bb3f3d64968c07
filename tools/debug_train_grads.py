"""Per-tensor gradient comparison: native train step vs fp32 CPU autograd on the oracle."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import numpy as np, torch
from scipy import ndimage
from oracle import unet_ref, metrics_ref
from interactive_unet.unet import UNet
from interactive_unet.train_engine import TrainEngine

def run(dim, shape, dtype, loss_scale):
    N, ncls = 2, 2
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = UNet(lr=1e-3, num_classes=ncls, dim=dim, act_dtype=dtype, pretrained=False)
    p0 = unet_ref.init_params(dim=dim, ncls=ncls, seed=5)
    model.load_named(p0); model = model.cuda()
    rng = np.random.default_rng(0)
    img = np.stack([ndimage.gaussian_filter(rng.random(shape), 2) for _ in range(N)])
    img = (255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)[:, None]
    lab = img[:, 0] > 127
    y = np.stack([~lab, lab], 1).astype(np.float32)
    wt = np.repeat((rng.random((N, 1) + shape) > 0.2).astype(np.float32), ncls, 1)
    y = y * wt
    X = torch.tensor(img.astype(np.float32) / 255.0)
    pr = {k: v.clone().requires_grad_(not unet_ref.is_buffer(k)) for k, v in p0.items()}
    logits = unet_ref.forward_logits(pr, X, dim=dim, training=True)
    probs = torch.softmax(logits, 1)
    axes = (0,) + tuple(range(2, 2 + dim))
    gp = torch.tensor(metrics_ref.loss_grad('mcc_ce', probs.detach().numpy(), y, wt, axes=axes)).float()
    probs.backward(gp)
    te = TrainEngine(model, lr=1e-3, loss_kind='mcc_ce', loss_scale=loss_scale)
    Xd, yd, wd, N_, D, H, W, vox, xs = te._prep(X, torch.tensor(y), torch.tensor(wt))
    ws = te.forward_train(Xd, xs, N_, D, H, W)
    tdt, wd = te.loss_forward(ws, ws['z.dec0.conv2'], yd, wd, N_, vox)
    te.backward(ws, Xd, xs, yd, wd, tdt, N_)
    torch.cuda.synchronize()
    print(f'--- {dim}-D {dtype} loss {ws["out4"][0].item():.5f}')
    for name in te.names:
        gn = te.g(name).cpu().reshape(pr[name].shape) / te.loss_scale
        gr = pr[name].grad
        cos = torch.nn.functional.cosine_similarity(gn.flatten(), gr.flatten(), dim=0).item()
        rel = ((gn - gr).norm() / (gr.norm() + 1e-20)).item()
        print(f'{name:28s} cos {cos:.5f} rel {rel:.4f} |g| {gr.norm().item():.3e} |gn| {gn.norm().item():.3e}')

if __name__ == '__main__':
    run(2, (64, 96), 'fp16', 256.0)
    run(3, (16, 32, 32), 'bf16', 1.0)
