"""CPU-side checks of the drop-in boundary: libiunet.so loads and exports every symbol
include/iunet.h declares; argument validation works without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from interactive_unet import _native
    if not os.path.isfile(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _native, ctypes.CDLL(_native.LIB_PATH)


def test_library_exports_every_declared_symbol():
    nv, lib = _lib()
    header = open(os.path.join(ROOT, 'include', 'iunet.h')).read()
    declared = set(re.findall(r'\b(iunet_[a-z0-9_A-Z]+)\s*\(', header))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/iunet.h but not exported'
    for name in nv.exported_symbols():
        assert name in declared, f'{name} bound by _native.py but not declared in include/iunet.h'


def test_error_reporting_without_gpu():
    nv, _ = _lib()
    l = nv.lib()
    assert l.iunet_abi_version() >= 1
    # argument validation happens before any HIP call: bad dtype -> error code + message
    rc = l.iunet_pack_conv3(7, None, None, None, 32, 32, 9, 0, None)
    assert rc < 0 and b'dtype' in l.iunet_last_error()
    with pytest.raises(nv.NativeError):
        nv.check(rc)


def test_zoom_table_host_function_matches_oracle():
    """iunet_zoom_nearest_table / _len are host-only arithmetic (no GPU): identical to the oracle's table (itself pinned
    against scipy.ndimage.zoom in test_oracle_golden.py), and bad arguments are refused."""
    import numpy as np
    from oracle import multiscale_ref as mr
    nv, _ = _lib()
    l = nv.lib()
    for n in list(range(1, 200)) + [255, 256, 257, 384, 512, 1000, 1024]:
        for zoom in (0.5, 0.25, 0.3, 0.75):
            want = mr.zoom_table(n, zoom)
            m = l.iunet_zoom_nearest_len(n, zoom)
            assert m == len(want), (n, zoom)
            if m == 0:
                continue
            t = (nv.c_int * m)()
            assert l.iunet_zoom_nearest_table(n, zoom, t, m) == 0
            assert np.array_equal(np.frombuffer(t, dtype=np.int32), want), (n, zoom)
    t = (nv.c_int * 4)()
    assert l.iunet_zoom_nearest_table(8, 0.5, t, 3) < 0 and b'n_out' in l.iunet_last_error()
    assert l.iunet_zoom_nearest_len(0, 0.5) == 0
