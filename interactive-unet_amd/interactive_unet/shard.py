"""Multi-GPU sharding of the whole-volume prediction (predict.py:201-256), one process per GPU.

The reference only sketches "one block per GPU" in comments (predict.py:137-147, :204-232).
Here:

* data ownership   z-slabs of the volume: rank r owns planes [r*h, (r+1)*h) of the uint8 input, of the float32
                    accumulators and of the uint8 output -- and NOTHING else: its accumulators are slab-sized
                    (1.6 GB of the 12 GiB at 1024^3 on 8 ranks);
* compute          the flat (i, j, k) block list is cut into W contiguous runs (balanced to one block; a split by
                    block planes would cap 8 GPUs at 5.5x for an 11^3 grid); a rank runs the network on its blocks
                    and keeps their probabilities [S,S,S,C] in HBM;
* input exchange   every rank receives exactly the planes its block run touches (its z footprint: the clipped blocks'
                    extent -- reflect padding mirrors inside a block's own range), point to point from their owners
                    into one preallocated window: <= 3 of 11 block planes per rank at 1024^3 / 8 ranks instead of the
                    whole volume;
* output exchange  a block's probabilities are cut along z at the slab boundaries into PIECES (contiguous memory);
                    the pieces that fall in another rank's slab go to that owner point to point (xGMI is point to
                    point: each peer has its own link), in `rounds` grouped exchanges issued while the next blocks
                    compute, into slices of one preallocated pool;
* blending         every owner blends the pieces of its slab -- its own and the received ones -- in FLAT BLOCK
                    ORDER, as far as they have arrived after every round (a rank computes the tail of its run first:
                    those pieces head the next owner's list), with the same kernel as the single-process path (pred += P * win, weight += win, each
                    operation separately rounded).  Every voxel therefore sees exactly the additions of the
                    one-process loop in exactly its order: the N-rank result is byte-identical to the 1-rank
                    result (tests/test_shard_cpu.py), which fp32 partial sums per rank cannot give.

The compute is behind a small `ops` interface so the same sharding logic runs on the GPU (NativeOps -> libiunet)
and in the world_size-2 / 3 gloo tests on CPU (a numpy stand-in there).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import predict as P


def partition_blocks(nblocks, world):
    """Contiguous, balanced split of the flat block list: [(lo, hi)] per rank."""
    return [(r * nblocks // world, (r + 1) * nblocks // world) for r in range(world)]


def slab_bounds(Z, world):
    """z-slab ownership, equal padded height so collectives see equal sizes."""
    h = -(-Z // world)
    return [(min(r * h, Z), min((r + 1) * h, Z)) for r in range(world)], h


def footprint(block_coords, lo, hi):
    """z-extent [z0, z1) touched by blocks lo..hi-1 (clipped volume coordinates)."""
    if hi <= lo:
        return 0, 0
    b = np.asarray(block_coords[lo:hi])
    return int(b[:, 0].min()), int(b[:, 3].max())


def block_pieces(block, local, bounds):
    """Cut one block along z at the slab boundaries: [(owner, za, zb, pa)] with [za, zb) the piece's planes in volume
    coordinates and pa the block-local z of its first plane (block = clipped volume coords, local = coords inside
    the S^3 block, as get_block_coordinates returns them)."""
    out = []
    for q, (s0, s1) in enumerate(bounds):
        za, zb = max(int(block[0]), s0), min(int(block[3]), s1)
        if zb > za:
            out.append((q, za, zb, int(local[0]) + za - int(block[0])))
    return out


class NativeOps:
    """GPU implementation of the per-block work (libiunet through the predict shim)."""

    def __init__(self, model, num_classes, input_size, batch_size=None, axes=(0, 1, 2)):
        self.model, self.C, self.S, self.bs, self.axes = model, num_classes, input_size, batch_size, list(axes)
        self.device = model.device
        self.eng = model.engine('eval')
        self._acc, self._store, self._blk = {}, None, None
        self.round_align = max(1, int(P.BLOCK_BATCH)) if self.eng.dim == 3 else 1

    def agree_form(self, comm, volume, padded_first):
        """Ranks in SEPARATE processes take one decision about the prediction form (engine_auto.EngineAuto: x2m or fp16x2): each runs
        the calibration on its own first block (none: contributes 0) and the figure is all-reduced (MAX) -- entered by every rank at the
        same weight-load counts whatever its local state, at most once per `recal_every` weight loads; blocking (one 8-byte read).
        In-process communicators (the virtual-rank tests) share one engine and need no agreement."""
        eng = self.eng
        if not isinstance(comm, DistComm) or comm.world <= 1 or not hasattr(eng, 'collective_due') or not eng.collective_due():
            return
        S = self.S
        x, xs, dims = None, None, (S, S, S)
        if padded_first is not None:
            if eng.dim == 3:
                x = P.gather_block(volume, padded_first, S)
                xs = (S ** 3, S ** 3, S * S, S, 1)
            else:
                x = P.gather_block(volume, padded_first, S)
                xs, dims = (S * S, 0, 0, S, 1), (1, S, S)          # the block's slices along axis 0 (predict.py:87-98)
        eng.calibrate(x, xs, dims[0], dims[1], dims[2], blocking=True, group=comm.group if comm.group is not None else True)

    # ---- single-rank path: blend at once into a whole-volume accumulator (predict.predict_volume_array's loop)
    def make_accumulator(self, V):
        """Accumulators are cached per shape (the Gaussian window and 12 B/voxel of HBM are not re-created for every
        volume of a series); a re-used one is zeroed."""
        acc = self._acc.get(tuple(V))
        if acc is None:
            acc = P.VolumeAccumulator(V, self.C, self.S, self.device)
            self._acc = {tuple(V): acc}
        else:
            acc.reset()
        return acc

    def predict_run(self, acc, volume, bc, pbc, lbc, lo, hi):
        """Blocks lo..hi-1 of the flat list, blended in list order (3-D net: batched forward; 2-D net: 2.5-D)."""
        if self.eng.dim == 3:
            P.predict_blocks_3d(self.eng, acc, volume, bc, pbc, lbc, lo, hi)
        else:
            blk = torch.empty((self.S,) * 3, dtype=torch.uint8, device=self.device)
            for i in range(lo, hi):
                P.gather_block(volume, pbc[i], self.S, out=blk)
                P.predict_block_device(self.model, blk, acc.block_probs, self.C, self.bs, self.axes)
                acc.blend(bc[i], lbc[i])

    def finalize(self, acc):
        return acc.finalize()

    # ---- multi-rank path: probabilities kept per block, blended later by the slab owners
    def new_store(self, n):
        """fp32 [n, S, S, S, C] probabilities of this rank's blocks (16 MB per 128^3 block at C = 2); cached."""
        S, C = self.S, self.C
        if self._store is None or self._store.shape[0] < n:
            self._store = torch.empty((max(n, 1), S, S, S, C), dtype=torch.float32, device=self.device)
        return self._store

    def new_window(self, shape):
        """uint8 planes of this rank's footprint (filled by the input exchange); cached."""
        w = getattr(self, '_window', None)
        if w is None or tuple(w.shape) != tuple(shape):
            w = self._window = torch.empty(tuple(shape), dtype=torch.uint8, device=self.device)
        return w

    def recv_pool(self, nplanes):
        """ONE fp32 buffer [nplanes, S, S, C] for every probability piece this rank receives (no allocation per piece); cached."""
        p = getattr(self, '_pool', None)
        if p is None or p.shape[0] < nplanes:
            p = self._pool = torch.empty((max(nplanes, 1), self.S, self.S, self.C), dtype=torch.float32, device=self.device)
        return p

    def hbm_bytes(self, acc=None):
        """Device bytes this rank's prediction holds: the footprint window, the block-probability store, the receive pool, the slab
        accumulators and the engine's activation workspaces + packed operators (whichever sequence -- Python or the C++ graph -- ran)."""
        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        total = nb(getattr(self, '_window', None)) + nb(self._store) + nb(getattr(self, '_pool', None))
        if acc is not None:
            total += sum(nb(getattr(acc, k, None)) for k in ('pred', 'weight', 'final', 'window', 'block_probs'))
        # (engine_auto.EngineAuto holds one engine per prediction form it has used: the calibration's second form counts too)
        for eng in (getattr(self.eng, '_engines', None) or {'': self.eng}).values():
            for ws in getattr(eng, '_ws_cache', {}).values():
                total += sum(nb(t) for k, t in ws.items() if torch.is_tensor(t))
            g = getattr(eng, '_g', None)
            if g is not None:
                total += sum(nb(t) for t in g._ws.values()) + nb(g.packed) + nb(g.flat)
        return total

    def forward_blocks(self, volume, padded, store, j0):
        """Probabilities of the blocks with padded coordinates `padded` into store[j0 : j0 + len(padded)]."""
        S, C, nb = self.S, self.C, len(padded)
        if self.eng.dim == 3:
            B = max(1, int(P.BLOCK_BATCH))
            if self._blk is None or self._blk.shape[0] < B:
                self._blk = torch.empty((B,) + (S,) * 3, dtype=torch.uint8, device=self.device)
            for i in range(0, nb, B):
                n = min(B, nb - i)
                for j in range(n):
                    P.gather_block(volume, padded[i + j], S, out=self._blk[j])
                self.eng.infer(self._blk, (S ** 3, S ** 3, S * S, S, 1), n, S, S, S, probs=store[j0 + i:j0 + i + n],
                               out_strides=(S ** 3 * C, 1, S * S * C, S * C, C))
        else:
            blk = torch.empty((S,) * 3, dtype=torch.uint8, device=self.device)
            for i in range(nb):
                P.gather_block(volume, padded[i], S, out=blk)
                P.predict_block_device(self.model, blk, store[j0 + i], C, self.bs, self.axes)

    def make_slab_accumulator(self, h, Y, X):
        acc = self._acc.get(('slab', h, Y, X))
        if acc is None:
            acc = P.VolumeAccumulator((h, Y, X), self.C, self.S, self.device)
            self._acc = {('slab', h, Y, X): acc}
        else:
            acc.reset()
        return acc

    def blend_piece(self, acc, piece, pa, block, local):
        """Blend planes [block[0], block[3]) (slab-relative) of one block into the slab accumulator; `piece` holds the
        block's probabilities from block-local plane `pa` on ([nz, S, S, C], contiguous)."""
        from . import _native as nv
        S, C = self.S, self.C
        base = piece.data_ptr() - pa * S * S * C * 4          # where block-local plane 0 would be; never dereferenced below pa
        nv.call('iunet_blend_accumulate', nv.ptr(acc.pred), nv.ptr(acc.weight), nv.c_void_p(base), nv.ptr(acc.window),
                acc.V[0], acc.V[1], acc.V[2], C, S, nv.int_array(block), nv.int_array(local), nv.stream())


class DistComm:
    """The communicator of the product path: torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, gloo
    in the CPU tests).  predict_volume_sharded needs exactly two operations of it."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_gather(self, t):
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return parts

    def exchange(self, sends, recvs):
        """One grouped point-to-point round (ncclGroupStart/End: no ordering deadlock between the pairs):
        sends [(tensor, dst)], recvs [(tensor, src)] -> work handles; the transfers run on RCCL's stream."""
        ops = [dist.P2POp(dist.isend, t, dst, group=self.group) for t, dst in sends]
        ops += [dist.P2POp(dist.irecv, t, src, group=self.group) for t, src in recvs]
        return dist.batch_isend_irecv(ops) if ops else []


def compute_order(run, pieces, rank):
    """The order in which rank `rank` runs the blocks of its contiguous run [lo, hi): first the blocks with a piece in a HIGHER
    rank's slab (the tail of the run: those pieces head that owner's flat-order blend list, so they must arrive first), then the
    rest, each group ascending.  Forward results do not depend on the order; the BLEND order stays the flat one."""
    lo, hi = run
    tail = [b for b in range(lo, hi) if any(q > rank for q, _, _, _ in pieces[b])]
    ts = set(tail)
    return tail + [b for b in range(lo, hi) if b not in ts]


def predict_volume_sharded(ops, my_slab, volume_shape, input_size, overlap=0.25, group=None, rounds=8, comm=None):
    """Whole-volume prediction across the ranks of `group` (or of `comm`, an object with DistComm's interface).

    my_slab: this rank's uint8 z-slab [h_r, Y, X] of the input (device of `ops`).
    Returns (uint8 [h_r, Y, X, C] result for the same slab, stats dict).  The result is byte-identical to the slab
    of the single-process result whatever the number of ranks.

    Multi-rank schedule (every rank derives the whole of it from the shapes alone):
    * input: a rank needs the planes its block run touches (its z footprint: <= 3 block planes of 11 at 1024^3 / 8 ranks; reflect
      padding mirrors inside a block's own clipped range) -- the owners send exactly those sub-slabs point to point into one
      preallocated window [f1 - f0, Y, X]; the blocks are gathered from the window with z shifted by f0;
    * rounds: `rounds` grouped exchanges of probability pieces, each issued behind the forwards that produce its pieces and in
      flight while the next round computes; receive buffers are slices of ONE preallocated pool;
    * blend: the owner walks its flat-order piece list as far as it has arrived after every round (own pieces: as soon as
      their forward is enqueued; received ones: one round later, after that round's handles are waited for on the stream), so
      what is left behind the last round is that round's pieces only."""
    comm = comm or DistComm(group)
    world, rank = comm.world, comm.rank
    V = tuple(int(v) for v in volume_shape)
    S = int(input_size)
    bounds, h = slab_bounds(V[0], world)
    dev = my_slab.device
    bc, pbc, lbc = P.get_block_coordinates(np.array(V), input_size=S, overlap=overlap)
    runs = partition_blocks(len(pbc), world)
    lo, hi = runs[rank]
    z0, z1 = bounds[rank]
    if world == 1:
        acc = ops.make_accumulator(V)
        ops.predict_run(acc, my_slab, bc, pbc, lbc, lo, hi)
        return ops.finalize(acc), {'blocks': hi - lo, 'bytes_sent': 0, 'slab': (z0, z1), 'pieces_blended': hi - lo,
                                   'rounds': 0}
    # ---- input exchange: every rank gets the planes of its footprint, nothing more ----
    foot = [footprint(bc, a, b) for a, b in runs]
    f0, f1 = foot[rank]
    window = ops.new_window((max(f1 - f0, 1),) + V[1:])
    sends, recvs, halo_recv = [], [], 0
    for q in range(world):
        a, b = max(z0, foot[q][0]), min(z1, foot[q][1])              # my planes that rank q needs
        if b > a and q != rank:
            sends.append((my_slab[a - z0:b - z0], q))
        a, b = max(bounds[q][0], f0), min(bounds[q][1], f1)          # rank q's planes that I need
        if b > a:
            if q == rank:
                window[a - f0:b - f0].copy_(my_slab[a - z0:b - z0])
            else:
                recvs.append((window[a - f0:b - f0], q))
                halo_recv += (b - a) * V[1] * V[2]
    for work in comm.exchange(sends, recvs):
        work.wait()
    shifted = pbc.copy()
    shifted[:, 0] -= f0
    shifted[:, 3] -= f0
    if hasattr(ops, 'agree_form'):
        ops.agree_form(comm, window, shifted[lo] if hi > lo else None)
    # ---- schedule (identical on every rank) ----
    pieces = [block_pieces(bc[b], lbc[b], bounds) for b in range(len(pbc))]
    orders = [compute_order(runs[r], pieces, r) for r in range(world)]
    longest = max(b - a for a, b in runs)
    rounds = max(1, min(int(rounds), longest))
    per_round = -(-longest // rounds)
    align = max(1, int(getattr(ops, 'round_align', 1)))         # whole forward batches per round (no odd block per round)
    per_round = -(-per_round // align) * align
    rounds = -(-longest // per_round)
    round_of, slot_of = {}, {}
    for r, order in enumerate(orders):
        for pos, b in enumerate(order):
            round_of[b] = pos // per_round
            if r == rank:
                slot_of[b] = pos
    # everything that lands in my slab, in flat block order: (block, source rank, pa, za, zb)
    mine = sorted((b, src, pa, za, zb) for src, order in enumerate(orders) for b in order
                  for q, za, zb, pa in pieces[b] if q == rank)
    pool = ops.recv_pool(sum(zb - za for b, src, pa, za, zb in mine if src != rank))
    buf_of, used = {}, 0
    for b, src, pa, za, zb in mine:
        if src != rank:
            buf_of[(b, za)] = pool[used:used + (zb - za)]
            used += zb - za
    store = ops.new_store(hi - lo)
    acc = ops.make_slab_accumulator(z1 - z0, V[1], V[2])
    nxt, sent, early = 0, 0, 0

    def blend_ready(done_round_remote, done_round_own):
        nonlocal nxt
        while nxt < len(mine):
            b, src, pa, za, zb = mine[nxt]
            if round_of[b] > (done_round_own if src == rank else done_round_remote):
                break
            piece = store[slot_of[b], pa:pa + (zb - za)] if src == rank else buf_of[(b, za)]
            c, l = bc[b], lbc[b]
            block = (za - z0, int(c[1]), int(c[2]), zb - z0, int(c[4]), int(c[5]))
            local = (pa, int(l[1]), int(l[2]), pa + (zb - za), int(l[4]), int(l[5]))
            ops.blend_piece(acc, piece, pa, block, local)
            nxt += 1

    handles = []
    for t in range(rounds):
        blocks_t = orders[rank][t * per_round:(t + 1) * per_round]
        if blocks_t:
            ops.forward_blocks(window, shifted[blocks_t], store, t * per_round)
        sends, recvs = [], []
        for src, order in enumerate(orders):
            for b in order[t * per_round:(t + 1) * per_round]:
                for q, za, zb, pa in pieces[b]:
                    if src == rank and q != rank:
                        piece = store[slot_of[b], pa:pa + (zb - za)]
                        sends.append((piece, q))
                        sent += piece.numel() * 4
                    elif src != rank and q == rank:
                        recvs.append((buf_of[(b, za)], src))
        handles.append(comm.exchange(sends, recvs))         # in flight while the next round's blocks compute
        if t > 0:
            for work in handles[t - 1]:                       # round t - 1 has had a whole round of compute to land
                work.wait()
        blend_ready(t - 1, t)
        if t == rounds - 2:
            early = nxt
    for work in handles[-1]:
        work.wait()
    blend_ready(rounds, rounds)
    assert nxt == len(mine)
    out = ops.finalize(acc)
    return out, {'blocks': hi - lo, 'bytes_sent': sent, 'slab': (z0, z1), 'pieces_blended': len(mine), 'rounds': rounds,
                 'halo_bytes_received': halo_recv, 'footprint': (f0, f1), 'pieces_blended_before_last_round': early,
                 'hbm_bytes': ops.hbm_bytes(acc) if hasattr(ops, 'hbm_bytes') else None}
