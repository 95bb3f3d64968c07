"""Zarr v3 store <-> HBM through zarr3.py (PCIe- and codec-inclusive, never part of bench.py's value): a V^3 uint8 volume with the
reference's layout (chunks 128^3 in shards 256^3) written from the device and read back.   python tools/bench_zarr.py [--size 512]"""
import argparse, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'interactive-unet_amd'))
import torch
from interactive_unet import zarr3
import bench

ap = argparse.ArgumentParser(); ap.add_argument('--size', type=int, default=512); ap.add_argument('--channels', type=int, default=0)
a = ap.parse_args()
V = a.size
vol = bench.synth_volume_slab(0, V, V, V, 'cuda')
if a.channels:
    vol = torch.stack([vol] + [255 - vol] * (a.channels - 1), -1).contiguous()
tmp = tempfile.mkdtemp(prefix='iunet_zarr_')
try:
    for comp in ('auto', None):
        root = zarr3.open(os.path.join(tmp, f'v_{comp}.zarr'), mode='w')
        extra = (a.channels,) if a.channels else ()
        arr = root.create_array(name='0', shape=tuple(vol.shape), chunks=(128,) * 3 + extra, shards=(256,) * 3 + extra, compressors=comp)
        torch.cuda.synchronize(); t0 = time.time()
        arr.from_device(vol)
        t1 = time.time()
        back = zarr3.open(os.path.join(tmp, f'v_{comp}.zarr'))['0'].to_device('cuda')
        torch.cuda.synchronize(); t2 = time.time()
        assert torch.equal(back, vol)
        disk = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(tmp, f'v_{comp}.zarr')) for f in fs)
        gb = vol.numel() / 1e9
        print(f'{tuple(vol.shape)} uint8, compressors={comp}: write {t1 - t0:.2f} s ({gb / (t1 - t0):.2f} GB/s), read {t2 - t1:.2f} s '
              f'({gb / (t2 - t1):.2f} GB/s), on disk {disk / 1e6:.0f} MB of {vol.numel() / 1e6:.0f} MB')
finally:
    shutil.rmtree(tmp, ignore_errors=True)
