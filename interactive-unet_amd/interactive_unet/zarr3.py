"""Zarr v3 block I/O for the volume predictor, without the `zarr` package.

The reference reads `data/image_volumes/<name>.zarr['0']` (uint8 [Z,Y,X]) and writes
`data/predicted_volumes/<name>.zarr['0']` (uint8 [Z,Y,X,C]) plus pyramid levels '1'..'n' through zarr-python 3.1.3
with one fixed layout: inner chunks of 128^3 inside shards of 256^3 (predict.py:168-199, :252-261; utils.py:29-98).
This module reads and writes exactly that family of arrays -- regular chunk grid, `sharding_indexed` codec with an
inner `bytes` codec and an optional `zstd` / `gzip` compressor, little-endian uint64 index with a crc32c trailer at
the end (or start) of the shard file -- following the Zarr v3 core spec and the sharding codec spec (ZEP 2).  It is
NOT a general Zarr implementation: other codecs, dtypes wider than one byte with big-endian storage, storage
transformers and v2 arrays raise.

Parity status: the reference holds no Zarr fixture and zarr-python is absent from this image, so the byte layout is
**unpinned at the zarr-python boundary**; it is pinned against hand-assembled, spec-derived shard files and against an
independent zstd implementation (tests/test_zarr3_cpu.py).

The subset of the zarr-python API the reference uses is mirrored (`open`, `Group.create_array`, `Group.array_keys`,
`group[name]`, `Array.shape / .chunks / .shards / .dtype`, basic-slice `__getitem__` / `__setitem__`), so the native
predict.py / multiscale.py read like the reference's.  On top of it, `Array.to_device` / `Array.from_device` move a
whole volume between the store and HBM shard by shard: worker threads decode / encode the inner chunks (libzstd
releases the GIL), a ring of pinned shard buffers carries them over PCIe asynchronously.
"""
import builtins
import ctypes
import ctypes.util
import json
import os
import shutil
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

MISSING = 2 ** 64 - 1


# --------------------------------------------------------------------------- crc32c (Castagnoli), table driven
def _crc32c_table():
    poly, tab = 0x82F63B78, []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data, crc=0):
    """CRC-32C (iSCSI) of `data`; used on the shard index only (<= a few KB), so plain Python is enough."""
    crc ^= 0xFFFFFFFF
    tab = _CRC_TAB
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


# --------------------------------------------------------------------------- zstd through the system's libzstd
class _Zstd:
    _lib = None
    _lock = threading.Lock()

    @classmethod
    def lib(cls):
        with cls._lock:
            if cls._lib is None:
                name = ctypes.util.find_library('zstd') or 'libzstd.so.1'
                try:
                    l = ctypes.CDLL(name)
                except OSError as e:
                    raise RuntimeError('zarr3: this array is zstd-compressed and libzstd was not found on the system') from e
                l.ZSTD_compressBound.restype = ctypes.c_size_t
                l.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
                l.ZSTD_compress.restype = ctypes.c_size_t
                l.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                l.ZSTD_decompress.restype = ctypes.c_size_t
                l.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
                l.ZSTD_isError.restype = ctypes.c_uint
                l.ZSTD_isError.argtypes = [ctypes.c_size_t]
                cls._lib = l
        return cls._lib


def zstd_compress(buf, level=0):
    """One zstd frame of `buf` (bytes-like / contiguous numpy array).  level 0 = libzstd's default, as numcodecs."""
    l = _Zstd.lib()
    src = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.reshape(-1).view(np.uint8)
    cap = l.ZSTD_compressBound(src.size)
    dst = np.empty(cap, dtype=np.uint8)
    n = l.ZSTD_compress(dst.ctypes.data, cap, src.ctypes.data, src.size, int(level))
    if l.ZSTD_isError(n):
        raise RuntimeError('zarr3: ZSTD_compress failed')
    return dst[:n].tobytes()


def zstd_decompress_into(src, out):
    """Decompress one frame into the contiguous uint8 numpy array `out`; its size must be the frame's content size."""
    l = _Zstd.lib()
    s = np.frombuffer(src, dtype=np.uint8)
    n = l.ZSTD_decompress(out.ctypes.data, out.size, s.ctypes.data, s.size)
    if l.ZSTD_isError(n) or n != out.size:
        raise ValueError(f'zarr3: corrupt zstd chunk (got {n} bytes, expected {out.size})')


# --------------------------------------------------------------------------- metadata
def _codec_name(c):
    return c['name'] if isinstance(c, dict) else c


def _ceil_div(a, b):
    return -(-a // b)


class Array:
    """One sharded (or plain chunked) Zarr v3 array on a directory store."""

    def __init__(self, path, meta, readonly=True):
        self.path, self.meta, self.readonly = path, meta, readonly
        self.shape = tuple(int(v) for v in meta['shape'])
        self.dtype = np.dtype(meta['data_type'])
        if self.dtype.itemsize != 1:
            raise NotImplementedError(f'zarr3: only one-byte dtypes (uint8 / int8 / bool), got {self.dtype}')
        if meta.get('storage_transformers'):
            raise NotImplementedError('zarr3: storage transformers are not supported')
        grid = meta['chunk_grid']
        if grid['name'] != 'regular':
            raise NotImplementedError(f"zarr3: chunk grid {grid['name']!r}")
        outer = tuple(int(v) for v in grid['configuration']['chunk_shape'])
        enc = meta.get('chunk_key_encoding', {'name': 'default', 'configuration': {'separator': '/'}})
        self._key_prefix = 'c' if enc['name'] == 'default' else ''
        self._sep = enc.get('configuration', {}).get('separator', '/' if enc['name'] == 'default' else '.')
        self.fill = np.array(meta.get('fill_value', 0) or 0).astype(self.dtype)
        codecs = meta['codecs']
        if len(codecs) == 1 and _codec_name(codecs[0]) == 'sharding_indexed':
            cfg = codecs[0]['configuration']
            self.shards, self.chunks = outer, tuple(int(v) for v in cfg['chunk_shape'])
            self._inner = cfg['codecs']
            idx = [_codec_name(c) for c in cfg.get('index_codecs', [{'name': 'bytes'}, {'name': 'crc32c'}])]
            if idx not in (['bytes'], ['bytes', 'crc32c']):
                raise NotImplementedError(f'zarr3: index codecs {idx}')
            self._index_crc = idx[-1] == 'crc32c'
            self._index_end = cfg.get('index_location', 'end') == 'end'
        else:                                             # unsharded: one "shard" = one chunk, no index
            self.shards, self.chunks, self._inner = None, outer, codecs
            self._index_crc, self._index_end = False, True
        names = [_codec_name(c) for c in self._inner]
        if not names or names[0] != 'bytes' or any(n not in ('bytes', 'zstd', 'gzip', 'crc32c') for n in names):
            raise NotImplementedError(f'zarr3: inner codec chain {names} (supported: bytes [+ zstd | gzip] [+ crc32c])')
        self._outer = self.shards or self.chunks
        if any(o % c for o, c in zip(self._outer, self.chunks)):
            raise ValueError('zarr3: the shard shape must be a multiple of the chunk shape')
        self._cps = tuple(o // c for o, c in zip(self._outer, self.chunks))      # inner chunks per shard, per axis
        self.ndim = len(self.shape)

    # ---- zarr-python surface
    @property
    def nshards(self):
        return tuple(_ceil_div(s, o) for s, o in zip(self.shape, self._outer))

    def _shard_file(self, sidx):
        parts = ([self._key_prefix] if self._key_prefix else []) + [str(i) for i in sidx]
        return os.path.join(self.path, *self._sep.join(parts).split('/'))

    # ---- inner chunk codec chain
    def _decode_chunk(self, raw, out):
        """`raw` bytes of one encoded inner chunk -> the contiguous uint8 view `out` (chunk-shaped)."""
        flat = out.reshape(-1)
        for c in reversed(self._inner[1:]):
            n = _codec_name(c)
            if n == 'crc32c':
                body, tail = raw[:-4], raw[-4:]
                if struct.unpack('<I', tail)[0] != crc32c(body):
                    raise ValueError('zarr3: chunk crc32c mismatch')
                raw = body
            elif n == 'zstd':
                zstd_decompress_into(raw, flat.view(np.uint8))
                return
            elif n == 'gzip':
                raw = zlib.decompress(raw, 16 + zlib.MAX_WBITS)
        if len(raw) != flat.size:
            raise ValueError(f'zarr3: chunk of {len(raw)} bytes, expected {flat.size}')
        flat.view(np.uint8)[:] = np.frombuffer(raw, dtype=np.uint8)

    def _encode_chunk(self, arr):
        raw = None
        for c in self._inner[1:]:
            n = _codec_name(c)
            cfg = c.get('configuration', {}) if isinstance(c, dict) else {}
            if n == 'zstd':
                raw = zstd_compress(arr if raw is None else raw, cfg.get('level', 0))
            elif n == 'gzip':
                co = zlib.compressobj(cfg.get('level', 5), zlib.DEFLATED, 16 + zlib.MAX_WBITS)
                raw = co.compress(arr.tobytes() if raw is None else raw) + co.flush()
            elif n == 'crc32c':
                raw = arr.tobytes() if raw is None else raw
                raw += struct.pack('<I', crc32c(raw))
        return arr.tobytes() if raw is None else raw

    # ---- one shard <-> one outer-chunk-shaped numpy array
    def read_shard(self, sidx, out=None, pool=None):
        """The shard with grid index `sidx` as a full shard-shaped array (fill value where nothing is stored).  pool: an executor
        that decodes the inner chunks side by side."""
        if out is None:
            out = np.empty(self._outer, dtype=self.dtype)
        f = self._shard_file(sidx)
        if not os.path.isfile(f):
            out[...] = self.fill
            return out
        with builtins.open(f, 'rb') as fh:
            data = fh.read()
        if self.shards is None:
            self._decode_chunk(data, out)
            return out
        n = int(np.prod(self._cps))
        isz = 16 * n + (4 if self._index_crc else 0)
        if len(data) < isz:
            raise ValueError(f'zarr3: shard {f} is shorter than its index')
        ibytes = data[-isz:] if self._index_end else data[:isz]
        if self._index_crc and struct.unpack('<I', ibytes[-4:])[0] != crc32c(ibytes[:-4]):
            raise ValueError(f'zarr3: shard index crc32c mismatch in {f}')
        index = np.frombuffer(ibytes[:16 * n], dtype='<u8').reshape(n, 2)
        todo = []
        for ci, cidx in enumerate(np.ndindex(*self._cps)):
            off, nb = int(index[ci, 0]), int(index[ci, 1])
            box = tuple(slice(i * c, (i + 1) * c) for i, c in zip(cidx, self.chunks))
            if off == MISSING and nb == MISSING:
                out[box] = self.fill
                continue
            if off + nb > len(data):
                raise ValueError(f'zarr3: chunk {cidx} of {f} points outside the file')
            todo.append((box, off, nb))

        def decode(item):
            box, off, nb = item
            tmp = np.empty(self.chunks, dtype=self.dtype)
            self._decode_chunk(data[off:off + nb], tmp)
            out[box] = tmp                                    # (the chunks of a shard are disjoint boxes of `out`)
        for _ in (pool.map(decode, todo) if pool is not None else map(decode, todo)):
            pass
        return out

    def write_shard(self, sidx, block, pool=None):
        """Store the shard-shaped array `block` as shard `sidx`.  Inner chunks that lie entirely beyond the array's
        bounds are not stored (index entry 2^64-1), the others are stored whole; an all-fill shard still gets a file.
        pool: an executor that encodes the inner chunks side by side (libzstd releases the interpreter lock)."""
        if self.readonly:
            raise PermissionError('zarr3: array opened read-only')
        f = self._shard_file(sidx)
        os.makedirs(os.path.dirname(f), exist_ok=True)
        if self.shards is None:
            payload = self._encode_chunk(np.ascontiguousarray(block))
        else:
            n = int(np.prod(self._cps))
            index = np.full((n, 2), MISSING, dtype='<u8')
            parts, pos = [], 0
            isz = 16 * n + (4 if self._index_crc else 0)
            base = 0 if self._index_end else isz
            todo = []
            for ci, cidx in enumerate(np.ndindex(*self._cps)):
                start = [s * o + i * c for s, o, i, c in zip(sidx, self._outer, cidx, self.chunks)]
                if any(st >= sh for st, sh in zip(start, self.shape)):
                    continue
                todo.append((ci, tuple(slice(i * c, (i + 1) * c) for i, c in zip(cidx, self.chunks))))
            encode = lambda item: self._encode_chunk(np.ascontiguousarray(block[item[1]]))
            for (ci, _), enc in zip(todo, pool.map(encode, todo) if pool is not None else map(encode, todo)):
                index[ci] = (base + pos, len(enc))
                parts.append(enc)
                pos += len(enc)
            ib = index.tobytes()
            if self._index_crc:
                ib += struct.pack('<I', crc32c(ib))
            payload = b''.join(parts + [ib]) if self._index_end else b''.join([ib] + parts)
        tmp = f + '.partial'
        with builtins.open(tmp, 'wb') as fh:
            fh.write(payload)
        os.replace(tmp, f)

    # ---- basic-slice access (what the reference does through zarr-python)
    def _norm(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        out = []
        for k, n in zip(key, self.shape):
            if not isinstance(k, slice) or k.step not in (None, 1):
                raise NotImplementedError('zarr3: only contiguous slices')
            a, b, _ = k.indices(n)
            out.append((a, max(a, b)))
        return out

    def __getitem__(self, key):
        reg = self._norm(key)
        res = np.empty([b - a for a, b in reg], dtype=self.dtype)
        lo = [a // o for (a, _), o in zip(reg, self._outer)]
        hi = [_ceil_div(b, o) if b > a else a // o for (a, b), o in zip(reg, self._outer)]
        buf = np.empty(self._outer, dtype=self.dtype)
        for sidx in np.ndindex(*[h - l for l, h in zip(lo, hi)]):
            sidx = tuple(s + l for s, l in zip(sidx, lo))
            self.read_shard(sidx, buf)
            src, dst = [], []
            for s, o, (a, b) in zip(sidx, self._outer, reg):
                g0, g1 = max(a, s * o), min(b, (s + 1) * o)
                src.append(slice(g0 - s * o, g1 - s * o))
                dst.append(slice(g0 - a, g1 - a))
            res[tuple(dst)] = buf[tuple(src)]
        return res

    def __setitem__(self, key, value):
        reg = self._norm(key)
        value = np.broadcast_to(np.asarray(value, dtype=self.dtype), [b - a for a, b in reg])
        lo = [a // o for (a, _), o in zip(reg, self._outer)]
        hi = [_ceil_div(b, o) for (_, b), o in zip(reg, self._outer)]
        for sidx in np.ndindex(*[max(0, h - l) for l, h in zip(lo, hi)]):
            sidx = tuple(s + l for s, l in zip(sidx, lo))
            src, dst, whole = [], [], True
            for s, o, (a, b), n in zip(sidx, self._outer, reg, self.shape):
                g0, g1 = max(a, s * o), min(b, (s + 1) * o)
                whole &= g0 == s * o and g1 == min((s + 1) * o, n)
                dst.append(slice(g0 - s * o, g1 - s * o))
                src.append(slice(g0 - a, g1 - a))
            buf = np.full(self._outer, self.fill, dtype=self.dtype) if whole else self.read_shard(sidx)
            buf[tuple(dst)] = value[tuple(src)]
            self.write_shard(sidx, buf)

    # ---- whole volume <-> HBM, shard by shard through pinned staging
    def to_device(self, device, out=None, workers=8, ring=4, decoders=None):
        """The whole array as a tensor on `device`: worker threads read shards into a ring of pinned shard buffers (their inner
        chunks decoded side by side by `decoders` more threads: the host cores of this process, at most 16), this thread issues one
        asynchronous host-to-device copy per shard."""
        import torch
        tdt = {np.dtype('uint8'): torch.uint8, np.dtype('int8'): torch.int8, np.dtype('bool'): torch.bool}[self.dtype]
        dev = torch.device(device)
        if out is None:
            out = torch.empty(self.shape, dtype=tdt, device=dev)
        pin = dev.type == 'cuda'
        bufs = [torch.empty(self._outer, dtype=tdt, pin_memory=pin) for _ in range(ring)]
        events = [None] * ring
        grid = list(np.ndindex(*self.nshards))

        if decoders is None:
            try:
                decoders = min(16, len(os.sched_getaffinity(0)))
            except AttributeError:
                decoders = min(16, os.cpu_count() or 1)

        def load(i):
            slot = i % ring
            if events[slot] is not None:
                events[slot].synchronize()                    # the copy that last used this buffer has finished
            self.read_shard(grid[i], bufs[slot].numpy(), pool=dec)
            return i
        with ThreadPoolExecutor(max_workers=max(1, decoders)) as dec, ThreadPoolExecutor(max_workers=min(workers, ring)) as ex:
            futs = {}
            nxt = 0
            for i in range(len(grid)):
                while nxt < len(grid) and nxt < i + ring:
                    futs[nxt] = ex.submit(load, nxt)
                    nxt += 1
                futs.pop(i).result()
                sidx, slot = grid[i], i % ring
                box = tuple(slice(s * o, min((s + 1) * o, n)) for s, o, n in zip(sidx, self._outer, self.shape))
                src = bufs[slot][tuple(slice(0, b.stop - b.start) for b in box)]
                out[box].copy_(src, non_blocking=True)
                if pin:
                    events[slot] = torch.cuda.Event()
                    events[slot].record()
        if pin:
            torch.cuda.current_stream().synchronize()
        return out

    def from_device(self, tensor, workers=8, ring=4, encoders=None):
        """Store a device (or host) tensor of the array's shape: one asynchronous device-to-host copy per shard into a
        ring of pinned buffers, worker threads write the shard files; the inner chunks of the shards in flight are encoded
        side by side by `encoders` more threads (default: the host cores of this process, at most 16) -- a 256^3 x 2 shard
        is 8 chunks of 4 MB, and zstd at ~0.3 GB/s per core was what a whole prediction waited for."""
        import torch
        if tuple(tensor.shape) != self.shape:
            raise ValueError(f'zarr3: tensor {tuple(tensor.shape)} does not match array {self.shape}')
        pin = tensor.is_cuda
        bufs = [torch.empty(self._outer, dtype=tensor.dtype, pin_memory=pin) for _ in range(ring)]
        busy = [None] * ring
        grid = list(np.ndindex(*self.nshards))

        if encoders is None:
            try:
                encoders = min(16, len(os.sched_getaffinity(0)))
            except AttributeError:
                encoders = min(16, os.cpu_count() or 1)

        def store(i, ev):
            if ev is not None:
                ev.synchronize()
            self.write_shard(grid[i], bufs[i % ring].numpy(), pool=enc)
        with ThreadPoolExecutor(max_workers=max(1, encoders)) as enc, ThreadPoolExecutor(max_workers=min(workers, ring)) as ex:
            for i, sidx in enumerate(grid):
                slot = i % ring
                if busy[slot] is not None:
                    busy[slot].result()                       # the shard that last used this buffer is on disk
                box = tuple(slice(s * o, min((s + 1) * o, n)) for s, o, n in zip(sidx, self._outer, self.shape))
                dst = bufs[slot]
                ext = tuple(slice(0, b.stop - b.start) for b in box)
                if any(e.stop != o for e, o in zip(ext, self._outer)):
                    dst.fill_(int(self.fill))
                dst[ext].copy_(tensor[box], non_blocking=True)
                ev = None
                if pin:
                    ev = torch.cuda.Event()
                    ev.record()
                busy[slot] = ex.submit(store, i, ev)
            for f in busy:
                if f is not None:
                    f.result()


class Group:
    def __init__(self, path, readonly):
        self.path, self.readonly = path, readonly

    def __getitem__(self, name):
        p = os.path.join(self.path, str(name))
        mf = os.path.join(p, 'zarr.json')
        if not os.path.isfile(mf):
            raise KeyError(name)
        with builtins.open(mf) as f:
            meta = json.load(f)
        if meta.get('zarr_format') != 3:
            raise NotImplementedError('zarr3: only zarr_format 3')
        return Array(p, meta, self.readonly) if meta['node_type'] == 'array' else Group(p, self.readonly)

    def __contains__(self, name):
        return os.path.isfile(os.path.join(self.path, str(name), 'zarr.json'))

    def array_keys(self):
        out = []
        for d in sorted(os.listdir(self.path)):
            mf = os.path.join(self.path, d, 'zarr.json')
            if os.path.isfile(mf):
                with builtins.open(mf) as f:
                    if json.load(f).get('node_type') == 'array':
                        out.append(d)
        return out

    def create_array(self, name, shape, dtype='uint8', chunks=None, shards=None, overwrite=False, fill_value=0,
                     compressors='auto'):
        """zarr-python's Group.create_array for this layout: `chunks` = inner chunk shape, `shards` = shard shape
        (None: unsharded); compressors 'auto' = zstd level 0 without checksum (zarr-python 3's default for numeric
        data), None = uncompressed."""
        if self.readonly:
            raise PermissionError('zarr3: group opened read-only')
        p = os.path.join(self.path, str(name))
        if os.path.exists(p):
            if not overwrite:
                raise FileExistsError(p)
            shutil.rmtree(p)
        os.makedirs(p)
        shape = tuple(int(v) for v in shape)
        chunks = tuple(int(v) for v in (chunks or shape))
        inner = [{'name': 'bytes'}]
        if compressors == 'auto':
            inner.append({'name': 'zstd', 'configuration': {'level': 0, 'checksum': False}})
        elif compressors:
            inner += list(compressors)
        if shards is not None:
            shards = tuple(int(v) for v in shards)
            codecs = [{'name': 'sharding_indexed', 'configuration': {
                'chunk_shape': list(chunks), 'codecs': inner,
                'index_codecs': [{'name': 'bytes', 'configuration': {'endian': 'little'}}, {'name': 'crc32c'}],
                'index_location': 'end'}}]
            outer = shards
        else:
            codecs, outer = inner, chunks
        meta = {'zarr_format': 3, 'node_type': 'array', 'shape': list(shape), 'data_type': np.dtype(dtype).name,
                'chunk_grid': {'name': 'regular', 'configuration': {'chunk_shape': list(outer)}},
                'chunk_key_encoding': {'name': 'default', 'configuration': {'separator': '/'}},
                'fill_value': int(fill_value), 'codecs': codecs, 'attributes': {}, 'storage_transformers': []}
        with builtins.open(os.path.join(p, 'zarr.json'), 'w') as f:
            json.dump(meta, f, indent=2)
        return Array(p, meta, readonly=False)


def open(path, mode='r'):            # noqa: A001 -- zarr-python's name; file I/O in this module uses builtins.open
    """zarr.open(path, mode) for a directory store holding a v3 group: 'r' read-only, 'r+' read / write an existing
    group, 'a' the same but created when missing, 'w' create (replacing what is there)."""
    if mode not in ('r', 'r+', 'a', 'w'):
        raise ValueError(f'zarr3.open: mode {mode!r}')
    mf = os.path.join(path, 'zarr.json')
    if mode == 'w' and os.path.exists(path):
        shutil.rmtree(path)
    if mode in ('w', 'a') and not os.path.isfile(mf):
        os.makedirs(path, exist_ok=True)
        with builtins.open(mf, 'w') as f:
            json.dump({'zarr_format': 3, 'node_type': 'group', 'attributes': {}}, f, indent=2)
    if not os.path.isfile(mf):
        raise FileNotFoundError(f'zarr3.open: no Zarr v3 group at {path}')
    with builtins.open(mf) as f:
        meta = json.load(f)
    if meta.get('zarr_format') != 3:
        raise NotImplementedError('zarr3: only zarr_format 3 stores')
    if meta.get('node_type') == 'array':
        return Array(path, meta, readonly=(mode == 'r'))
    return Group(path, readonly=(mode == 'r'))

