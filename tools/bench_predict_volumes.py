"""predict.predict_volumes end to end (Zarr v3 store -> HBM -> tiled prediction -> Zarr v3 store + pyramid; PCIe- and codec-inclusive, never
part of bench.py's value): n synthetic V^3 uint8 volumes in a scratch data/ directory, 3-D U-Net (4 levels, base 32) in its default
prediction mode (fp16x2) or bf16.  Prints the wall time per volume and the parts measured alone (read, predict, encode + write).
python tools/bench_predict_volumes.py [n] [V] [fp16x2|bf16]"""
import os, shutil, sys, tempfile, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import numpy as np
import torch
import bench
from interactive_unet import predict, multiscale, zarr3
from interactive_unet.unet import UNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
V = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else 'fp16x2'
tmp = tempfile.mkdtemp(prefix='iunet_pv_')
cwd = os.getcwd()
try:
    os.chdir(tmp)
    os.makedirs('model'); os.makedirs('data/image_volumes'); os.makedirs('data/predicted_volumes')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = UNet(num_classes=2, dim=3, act_dtype='bf16', infer_dtype='fp16x2' if mode == 'fp16x2' else 'bf16', pretrained=False)
    m.reset_parameters(seed=0)
    m.save_checkpoint('model/model.ckpt')
    for i in range(n):
        vol = torch.roll(bench.synth_volume_slab(0, V, V, V, 'cuda'), 37 * i, 0).cpu().numpy()      # (odd extents would run into the reference's zoom quirk)
        multiscale.create_multiscale_zarr(vol, f'data/image_volumes/v{i}.zarr', chunk_size=128, shard_size=256)
    # the parts alone, on volume 0
    model = predict._load_model(1, 2, torch.device('cuda'))
    t0 = time.time(); volume = zarr3.open('data/image_volumes/v0.zarr')['0'].to_device('cuda'); torch.cuda.synchronize(); t_read = time.time() - t0
    predict.predict_volume_array(model, volume, 128, 2); torch.cuda.synchronize()
    t0 = time.time(); final = predict.predict_volume_array(model, volume, 128, 2); torch.cuda.synchronize(); t_pred = time.time() - t0
    root = zarr3.open('data/predicted_volumes/scratch.zarr', mode='w')
    arr = root.create_array(name='0', shape=list(final.shape), dtype='uint8', overwrite=True, chunks=(128,) * 3 + (2,), shards=(256,) * 3 + (2,))
    t0 = time.time(); arr.from_device(final); multiscale.add_multiscales('data/predicted_volumes/scratch.zarr', scale=0.5, level0=final); t_write = time.time() - t0
    shutil.rmtree('data/predicted_volumes/scratch.zarr')
    del final, volume
    t0 = time.time()
    predict.predict_volumes(input_size=128, num_classes=2)
    wall = time.time() - t0
    print(f'{n} x {V}^3 ({mode}): predict_volumes {wall:.2f} s = {wall / n:.2f} s per volume; alone: read {t_read:.2f} s, predict {t_pred:.2f} s, '
          f'encode + write + pyramid {t_write:.2f} s (sum {t_read + t_pred + t_write:.2f} s)')
finally:
    os.chdir(cwd)
    shutil.rmtree(tmp, ignore_errors=True)
