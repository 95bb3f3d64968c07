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
        self.pred = np.zeros(self.V + (C,), dtype=np.float32)
        self.weight = np.zeros(self.V, dtype=np.float32)
        self.window = predict_ref.gaussian_3d(S)


class NumpyOps:
    """CPU stand-in for shard.NativeOps: same interface, numpy arithmetic with the single-process loop's operations
    (product and sum separately rounded in float32, as predict.py:244-245 on float32 arrays)."""

    def __init__(self, C, S):
        self.C, self.S = C, S

    # single-rank path
    def make_accumulator(self, V):
        return _Acc(V, self.C, self.S)

    def predict_run(self, acc, volume, bc, pbc, lbc, lo, hi):
        for i in range(lo, hi):
            Pb = _stub_probs(predict_ref.get_padded_block(volume.numpy(), *pbc[i]), self.C)
            self._blend(acc, Pb, bc[i], lbc[i])

    def _blend(self, acc, Pb, block, local):
        i0, j0, k0, i1, j1, k1 = [int(v) for v in block]
        a0, b0, c0, a1, b1, c1 = [int(v) for v in local]
        w = acc.window[a0:a1, b0:b1, c0:c1]
        acc.pred[i0:i1, j0:j1, k0:k1] += Pb[a0:a1, b0:b1, c0:c1] * w[..., None]
        acc.weight[i0:i1, j0:j1, k0:k1] += w

    def finalize(self, acc):
        return torch.from_numpy((255 * acc.pred / np.maximum(acc.weight, 1e-3)[..., None]).astype('uint8'))

    # multi-rank path
    def new_store(self, n):
        return torch.zeros((max(n, 1),) + (self.S,) * 3 + (self.C,), dtype=torch.float32)

    def new_window(self, shape):
        return torch.zeros(tuple(shape), dtype=torch.uint8)

    def recv_pool(self, nplanes):
        return torch.zeros((max(nplanes, 1), self.S, self.S, self.C), dtype=torch.float32)

    def forward_blocks(self, volume, padded, store, j0):
        for i, pb in enumerate(padded):
            store[j0 + i] = torch.from_numpy(_stub_probs(predict_ref.get_padded_block(volume.numpy(), *pb), self.C))

    def make_slab_accumulator(self, h, Y, X):
        return _Acc((h, Y, X), self.C, self.S)

    def blend_piece(self, acc, piece, pa, block, local):
        full = np.zeros((self.S,) * 3 + (self.C,), dtype=np.float32)
        full[pa:pa + piece.shape[0]] = piece.numpy()
        self._blend(acc, full, block, local)


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


@pytest.mark.parametrize('world,V', [(2, (72, 56, 40)), (3, (50, 40, 33)), (3, (100, 33, 40))])
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
    # every voxel sees the additions of the one-process loop in its order (the owner blends the pieces of its slab in
    # flat block order): byte-identical, whatever the number of ranks
    assert np.array_equal(got, want)
    assert np.array_equal(got.argmax(-1), want.argmax(-1))


def test_partition_is_balanced_and_complete():
    from interactive_unet import shard
    for n, w in ((1331, 8), (125, 8), (27, 4), (5, 8)):
        runs = shard.partition_blocks(n, w)
        assert runs[0][0] == 0 and runs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
        sizes = [b - a for a, b in runs]
        assert max(sizes) - min(sizes) <= 1
    assert max(b - a for a, b in shard.partition_blocks(1331, 8)) == 167      # 99.6 % balance


# --------------------------------------------------------------------------- data-parallel training plumbing (dp.py)
def _dp_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from interactive_unet import dp
    g = torch.Generator().manual_seed(100 + rank)              # every rank draws ITS OWN initialisation, as UNet() does
    n, split = 1000, 640
    flat = torch.randn(n, generator=g)
    bufs = [torch.randn(7, generator=g), torch.rand(7, generator=g)]
    dp.broadcast_state(flat, bufs, dist.group.WORLD)
    start = flat.clone()
    grad = torch.zeros(n)
    buckets = dp.GradBuckets(grad, split, dist.group.WORLD)
    for step in range(2):                                      # two "training steps": local gradients differ per rank
        local = torch.randn(n, generator=g)
        grad[split:] = local[split:]                           # decoder + head gradients are ready first ...
        buckets.start_tail()                                   # ... and go out while the encoder part is "computed"
        if step == 0:                                          # the bottom encoder level is next: a bucket of its own
            grad[300:split] = local[300:split]                 # (step 1 goes without it: the two-bucket form of a net whose
            buckets.start(300, split)                          #  bottom level is not one run of the flat tensor)
            grad[:300] = local[:300]
        else:
            grad[:split] = local[:split]
        w = buckets.finish()                                   # reduces what is left ([0, 300)) and waits for all three
        flat -= 0.1 * grad / w
        q.put(('grad', rank, step, local.numpy().copy(), grad.numpy().copy()))
    q.put(('flat', rank, start.numpy(), flat.numpy().copy(), [b.numpy().copy() for b in bufs]))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_broadcast_and_bucketed_allreduce_world2():
    """without a parameter broadcast the ranks start from different weights and diverge silently.  After
    broadcast_state + two bucketed all-reduce steps both ranks hold bit-identical parameters, equal to rank 0's start
    minus the averaged gradients."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=120) for _ in range(world * 3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flats = {m[1]: m for m in msgs if m[0] == 'flat'}
    grads = {(m[1], m[2]): m for m in msgs if m[0] == 'grad'}
    assert np.array_equal(flats[0][2], flats[1][2])            # same start (rank 0's draw)
    assert np.array_equal(flats[0][3], flats[1][3])            # same parameters after two steps, bit for bit
    for a, b in zip(flats[0][4], flats[1][4]):
        assert np.array_equal(a, b)                            # BatchNorm running statistics too
    want = flats[0][2].copy()
    for step in range(2):
        total = grads[(0, step)][3] + grads[(1, step)][3]
        assert np.array_equal(grads[(0, step)][4], total) and np.array_equal(grads[(1, step)][4], total)
        want -= (0.1 * torch.from_numpy(total) / world).numpy()
    assert np.array_equal(flats[0][3], want)


# --------------------------------------------------------------------------- one prediction form on every rank (engine_auto.py, shard.agree_form)
def _agree_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from interactive_unet.engine_auto import EngineAuto
    e = EngineAuto(dim=3, device='cpu', recal_every=4)

    class _Form:                                   # stands in for the two EngineX2 forms: nothing is launched on a CPU-only rank
        def load_eval(self, params):
            pass
    e._form = lambda name: _Form()
    # what a rank's own tile measures in each agreement round: below the threshold on some ranks and above it on others
    script = {0: [1e-4, 1e-4, 3e-4], 1: [1e-4, 9e-4, 2e-4], 2: [None, None, None]}[rank]       # rank 2 has no block (contributes 0)
    it = iter(script)

    def measure(x, xs, D, H, W):
        v = next(it)
        return torch.tensor([0.0 if v is None else v, 5.0]), ((64, 64, 64) if v is not None else None)
    e._measure = measure
    log = []
    for load in range(1, 11):                      # ten weight loads = a data-parallel loop that predicts after every step
        e.load_eval({})
        if rank == 1 and load == 3:
            e.widen()                              # a rank-local event (its activations saturated): must not desynchronise the collective
        if e.collective_due():
            x = None if rank == 2 else object()
            e.calibrate(x, None, 64, 64, 64, blocking=True, group=True)
            log.append((load, e.mode, round(e.calibration['diff'], 6)))
    q.put((rank, log, e.form))
    dist.barrier()
    dist.destroy_process_group()


def test_prediction_form_is_agreed_over_the_process_group_world3():
    """Ranks of a sharded prediction calibrate on their OWN data and must still select ONE form (x2m or fp16x2): the figure is all-reduced
    (MAX), the collective is entered at the same weight-load counts by every rank whatever its local state (a widened rank, a rank without
    blocks), and a figure above the threshold on ANY rank sends every rank to fp16x2."""
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (log, form) for r, log, form in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(1, 'x2m', 1e-4), (5, 'fp16x2', 9e-4), (9, 'x2m', 3e-4)]                 # loads 1, 5, 9 (every 4); the group maximum decides
    for r in range(world):
        assert [(l, m, pytest.approx(d, rel=1e-3)) for l, m, d in want] == res[r][0], (r, res[r])
    assert res[0][1] == 'x2m' and res[2][1] == 'x2m' and res[1][1] == 'fp16x2_wide'   # rank 1 keeps its wider form, and kept taking part
