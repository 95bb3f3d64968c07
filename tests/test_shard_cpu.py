"""world_size-2 (and 3) gloo tests of the multi-GPU sharding logic on CPU: block-run partition,
slab all-gather, point-to-point accumulator exchange.  The per-block compute is a numpy
stand-in (oracle helpers) -- the sharding code is the product's (interactive_unet/shard.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import predict_ref


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_probs(blk, C):
    x = blk.astype(np.float32) / 255.0
    S = x.shape[0]
    r = (np.arange(S, dtype=np.float32) / S).reshape(S, 1, 1)
    l = np.stack([x * (1 + r), 0.7 - x, 0.2 + 0 * x][:C], -1)
    e = np.exp(l - l.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


class _Acc:
    def __init__(self, V, C, S):
        self.V, self.C, self.S = tuple(V), C, S
        self.pred = torch.zeros(self.V + (C,), dtype=torch.float32)
        self.weight = torch.zeros(self.V, dtype=torch.float32)
        self.window = predict_ref.gaussian_3d(S)


class NumpyOps:
    def __init__(self, C, S):
        self.C, self.S = C, S

    def make_accumulator(self, V):
        return _Acc(V, self.C, self.S)

    def predict_into(self, acc, volume, block, padded, local):
        blk = predict_ref.get_padded_block(volume.numpy(), *padded)
        Pb = _stub_probs(blk, self.C)
        i0, j0, k0, i1, j1, k1 = block
        a0, b0, c0, a1, b1, c1 = local
        w = acc.window[a0:a1, b0:b1, c0:c1]
        acc.pred[i0:i1, j0:j1, k0:k1] += torch.from_numpy(Pb[a0:a1, b0:b1, c0:c1] * w[..., None])
        acc.weight[i0:i1, j0:j1, k0:k1] += torch.from_numpy(w)

    def finalize_slab(self, acc, z0, z1):
        p, w = acc.pred[z0:z1].numpy(), acc.weight[z0:z1].numpy()
        return torch.from_numpy((255 * p / np.maximum(w, 1e-3)[..., None]).astype('uint8'))


def _worker(rank, world, port, V, S, C, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from interactive_unet import shard
    rng = np.random.default_rng(7)
    volume = rng.integers(0, 256, size=V, dtype=np.uint8)
    bounds, _ = shard.slab_bounds(V[0], world)
    z0, z1 = bounds[rank]
    out, stats = shard.predict_volume_sharded(NumpyOps(C, S), torch.from_numpy(volume[z0:z1].copy()), V, S, 0.25)
    q.put((rank, z0, z1, out.numpy(), stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,V', [(2, (72, 56, 40)), (3, (50, 40, 33))])
def test_sharded_predict_equals_single_process(world, V):
    S, C = 32, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, S, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(7)
    volume = rng.integers(0, 256, size=V, dtype=np.uint8)
    want, _, _ = predict_ref.blend_volume(volume, lambda b: _stub_probs((b * 255).round().astype(np.uint8), C), S, C)
    got = np.zeros(V + (C,), np.uint8)
    nblocks = 0
    for rank, z0, z1, out, stats in res:
        got[z0:z1] = out
        nblocks += stats['blocks']
        assert stats['bytes_sent'] > 0 or world == 1
    assert nblocks == len(predict_ref.get_block_coordinates(V, S, 0.25)[0])
    d = np.abs(got.astype(int) - want.astype(int))
    # partial sums are associated per rank, so a value sitting exactly on an integer may move by one LSB
    assert d.max() <= 1 and (d > 0).mean() < 1e-4, (d.max(), (d > 0).mean())
    assert np.array_equal(got.argmax(-1)[d.max(-1) == 0], want.argmax(-1)[d.max(-1) == 0])


def test_partition_is_balanced_and_complete():
    from interactive_unet import shard
    for n, w in ((1331, 8), (125, 8), (27, 4), (5, 8)):
        runs = shard.partition_blocks(n, w)
        assert runs[0][0] == 0 and runs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
        sizes = [b - a for a, b in runs]
        assert max(sizes) - min(sizes) <= 1
    assert max(b - a for a, b in shard.partition_blocks(1331, 8)) == 167      # 99.6 % balance
