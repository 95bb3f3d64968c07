"""Install the native hot path OVER a checkout of laprade117/interactive-unet.

    python tools/install_overlay.py /path/to/interactive-unet/interactive_unet [--symlink]

The reference's app.py does `from . import utils, trainer, predict, suggestor` (app.py:19-21): its modules resolve
inside ITS package directory, so a directory on PYTHONPATH is never consulted.  The drop-in is therefore file-level:
the hot-path modules of the reference (unet, trainer, predict, metrics, slicer, loader, suggestor) are replaced by the
native ones of the same names, the native-only modules (engine, engine_f32, engine_x2, engine_auto, net_graph, train_engine, train_engine_f32, shard, dp, multiscale, zarr3,
_native) are added beside them, and libiunet.so goes to <package>/../lib/ where _native.py looks for it.  The reference's
app.py, annotator.py, volumedata.py and -- deliberately -- utils.py are NOT touched: app.py:33-788 calls ~15 project /
TIFF / plotting helpers of utils.py that are outside the hot path.  The replaced files are kept as <name>.py.reference.
"""
import argparse
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), 'interactive-unet_amd')
REPLACED = ['unet', 'trainer', 'predict', 'metrics', 'slicer', 'loader', 'suggestor']
ADDED = ['_native', 'engine', 'engine_f32', 'engine_x2', 'engine_auto', 'net_graph', 'train_engine', 'train_engine_f32', 'shard', 'dp', 'multiscale', 'zarr3']
NOT_INSTALLED = ['utils', '__init__']          # the reference's own stay


def install(target, symlink=False, keep_backup=True):
    src = os.path.join(PKG, 'interactive_unet')
    if not os.path.isdir(target):
        raise SystemExit(f'{target} is not a directory')
    done = []
    for name in REPLACED + ADDED:
        s, d = os.path.join(src, name + '.py'), os.path.join(target, name + '.py')
        if not os.path.isfile(s):
            continue
        if os.path.lexists(d):
            if keep_backup and name in REPLACED and not os.path.lexists(d + '.reference'):
                os.replace(d, d + '.reference')
            else:
                os.remove(d)
        (os.symlink if symlink else shutil.copyfile)(s, d)
        done.append(name)
    lib_src = os.path.join(PKG, 'lib', 'libiunet.so')
    lib_dir = os.path.join(os.path.dirname(os.path.abspath(target)), 'lib')
    if os.path.isfile(lib_src):
        os.makedirs(lib_dir, exist_ok=True)
        d = os.path.join(lib_dir, 'libiunet.so')
        if os.path.lexists(d):
            os.remove(d)
        (os.symlink if symlink else shutil.copyfile)(lib_src, d)
    return done


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('target', help="the reference checkout's interactive_unet/ package directory")
    ap.add_argument('--symlink', action='store_true')
    a = ap.parse_args()
    print('installed:', ', '.join(install(a.target, a.symlink)))
    print('kept (reference):', ', '.join(NOT_INSTALLED + ['app', 'annotator', 'volumedata']))
    sys.exit(0)
