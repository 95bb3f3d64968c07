"""Native training step of the canonical U-Net: forward with BatchNorm batch statistics,
fused head + softmax + reference loss (metrics.py), full backward and AdamW, all on
libiunet kernels.  Replaces what Lightning + autograd + torch.optim.AdamW do under
unet.py:88-102 / trainer.py:56-63 ('16-mixed': fp32 master weights, fp16/bf16 compute).

PyTorch tensors are only device memory here; every op is a libiunet launch on the current
HIP stream, except the optional RCCL gradient all-reduce (torch.distributed).
"""
import ctypes
import math

import os

import torch

from . import _native as nv
from .engine import BN_EPS

LOSS_KINDS = {'ce': 0, 'dice': 1, 'iou': 2, 'mcc': 3, 'dice_ce': 4, 'iou_ce': 5, 'mcc_ce': 6}
BN_MOMENTUM = 0.1


def _vox(d):
    return d[0] * d[1] * d[2]


class TrainEngine:
    def __init__(self, model, lr=None, loss_kind='mcc_ce', betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 loss_scale=None, process_group=None):
        self.model = model
        self.dev = model.device
        if self.dev.type != 'cuda':
            raise RuntimeError('native training runs on the GPU only (no CPU fallback)')
        nv.lib()
        self.dim, self.levels, self.ch = model.dim, model.levels, [model.base * 2 ** l for l in range(model.levels)]
        self.cin, self.ncls = model.num_channels, model.num_classes
        self.T = model.act_dtype
        if self.T not in nv.DTYPE_CODE:
            raise NotImplementedError("native training runs with 16-bit activations (act_dtype 'fp16' / 'bf16'); "
                                      "act_dtype='fp32' is the inference parity mode")
        self.dt = nv.DTYPE_CODE[self.T]
        self.es = 2
        self.taps, self.npos = 3 ** self.dim, 2 ** self.dim
        self.lr = model.lr if lr is None else lr
        self.kind = LOSS_KINDS[loss_kind] if isinstance(loss_kind, str) else int(loss_kind)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        # loss scale, step count and the overflow back-off live ON THE DEVICE (csrc/train_pointwise.hip: training state): the head's
        # backward reads the scale there, the optimiser step checks the gradient, skips / halves / counts there -- no host read per step
        # (the fp16 step used to read its overflow flag on the host every step)
        self.dynamic_scale = self.T == torch.float16 and loss_scale is None
        self.state = torch.zeros(8, dtype=torch.float32, device=self.dev)
        nv.call('iunet_train_state_init', nv.ptr(self.state), float(loss_scale) if loss_scale is not None else (1024.0 if self.T == torch.float16 else 1.0),
                int(self.dynamic_scale), nv.stream())
        self.pg = process_group
        # the BatchNorm + ReLU between the two convs of a stage is applied by the consumers while they stage their input
        # wherever the second conv runs on layout 2 (3-D: always; 2-D: up to 64 channels) -- see _conv2_input
        # (IUNET_NO_ACT_FUSION=1: materialise it, for A/B runs)
        self.fuse_act = not os.environ.get('IUNET_NO_ACT_FUSION')
        # GroupNorm variant: statistics per (sample, group) -- the per-channel fusions of the BatchNorm path (activation applied in
        # the next conv's loaders, pooling folded into the norm passes, first-layer weight gradient with the norm backward inside)
        # do not apply; every norm is iunet_gn_relu_fwd / _bwd on materialised tensors
        self.norm, self.groups = getattr(model, 'norm', 'batch'), getattr(model, 'groups', 8)
        self.gn = self.norm == 'group'
        if self.gn:
            self.fuse_act = False
        # the BatchNorm-backward sums of a stage's conv1 ride in the epilogue of conv2's data gradient (IUNET_NO_BW_FUSION=1: separate
        # reduction pass, for A/B runs)
        self.fuse_bw = not self.gn and not os.environ.get('IUNET_NO_BW_FUSION')
        self.gn_conv_stats = self.gn and not os.environ.get('IUNET_NO_GN_CONV_STATS')
        self.gn_bw = self.gn_conv_stats and not os.environ.get('IUNET_NO_GN_BW_FUSION')      # the backward's sums from the data gradient's epilogue, per sample
        self.head_act = not self.gn and not os.environ.get('IUNET_NO_HEAD_ACT')     # A/B switch: materialise the last activation
        # head backward + the last conv's BatchNorm backward in two passes over its raw output (iunet_head_bn_bwd; IUNET_NO_HEAD_BN_FUSION=1:
        # the three-kernel sequence with the head's input gradient written and read back)
        self.head_bn = self.head_act and not os.environ.get('IUNET_NO_HEAD_BN_FUSION')
        # GroupNorm: the head reads the last conv's raw output with per-sample rows and its backward runs as iunet_head_gn_bwd (the last
        # activation and the head's input gradient are never written) -- only in that fused form (32 / 64 head channels, 2..4 classes)
        self.gn_head = (self.gn and not os.environ.get('IUNET_NO_HEAD_ACT') and not os.environ.get('IUNET_NO_HEAD_BN_FUSION')
                        and bool(nv.lib().iunet_head_bn_bwd_ok(self.ch[0], self.ncls)))
        self._bw_ready = {}
        self._flatten()
        if self.pg is not None:
            # every rank continues from rank 0's weights and BatchNorm statistics (each rank's module drew its own
            # initialisation, or read a checkpoint that rank 0 is about to replace)
            from . import dp
            dp.broadcast_state(self.flat, [model.tensor(n) for n in model._names
                                           if n.endswith('running_mean') or n.endswith('running_var')], self.pg)
            self.buckets = dp.GradBuckets(self.grad, self._dec_start, self.pg)
        self._alloc_packed()
        self._ws = {}
        self.probe = None          # timing hook of one layer's forward launches (bench.py roofline), see engine.Engine.probe
        self.use_handle = True     # False: every step sequenced from Python (tests compare the two)
        self.repack()
        model._packed_sig = None

    # ------------------------------------------------------------------ training state (device-resident; reading it synchronises)
    @property
    def loss_scale(self):
        return float(self.state[0].item())

    @loss_scale.setter
    def loss_scale(self, value):
        self.state[0:1].fill_(float(value))
        self._fixed_scale = None

    @property
    def step_count(self):
        return int(self.state.view(torch.int32)[1].item())

    @property
    def good_steps(self):
        return int(self.state.view(torch.int32)[2].item())

    @property
    def last_step_ok(self):
        """False when the last optimiser step was skipped because the fp16 gradient overflowed (a host read)."""
        return int(self.state.view(torch.int32)[3].item()) == 0

    # ------------------------------------------------------------------ parameters
    def _flatten(self):
        """Re-home every trainable parameter as a view of one flat fp32 master tensor
        (single AdamW launch, single all-reduce); gradients get the same layout."""
        m = self.model
        names = [n for n in m._names if not (n.endswith('running_mean') or n.endswith('running_var'))]
        sizes = [m.tensor(n).numel() for n in names]
        total = sum(sizes)
        flat = torch.empty(total, dtype=torch.float32, device=self.dev)
        off = 0
        self.offsets = {}
        for n, s in zip(names, sizes):
            t = m.tensor(n)
            flat[off:off + s].copy_(t.detach().reshape(-1))
            t.data = flat[off:off + s].view(t.shape)
            self.offsets[n] = (off, s)
            off += s
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.names = names
        # first element of the decoder + head parameters (they follow the encoder in the flat order): bucket boundary of
        # the data-parallel all-reduce
        first_dec = f'dec{self.levels - 2}.up.weight'
        self._dec_start = self.offsets[first_dec][0] if first_dec in self.offsets else 0
        # the bottom encoder level's parameters (its two convs hold almost half of the net): a bucket of their own, reduced while
        # the upper encoder levels still run their backward -- only if they are one run that ends where the decoder starts
        bottom = [self.offsets[n] for n in names if n.startswith(f'enc{self.levels - 1}.')]
        lo = min((o for o, _ in bottom), default=0)
        self._bottom_start = lo if bottom and lo + sum(s for _, s in bottom) == self._dec_start else None

    def p(self, name):
        return self.model.tensor(name)

    def g(self, name):
        off, s = self.offsets[name]
        return self.grad[off:off + s]

    def stage_names(self):
        return [f'enc{l}' for l in range(self.levels)] + [f'dec{l}' for l in range(self.levels - 2, -1, -1)]

    def stage_io(self, prefix):
        l = int(prefix[3:])
        ci = (self.cin if l == 0 else self.ch[l - 1]) if prefix.startswith('enc') else 2 * self.ch[l]
        return ci, self.ch[l], l

    def _alloc_packed(self):
        self.pk = {}
        for prefix in self.stage_names():
            ci, co, _ = self.stage_io(prefix)
            for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                name = f'{prefix}.conv{j}'
                if name == 'enc0.conv1':
                    self.pk[name] = (torch.empty(nv.lib().iunet_pack_first_conv_elems(b, a, self.taps), dtype=self.T,
                                                 device=self.dev), None)
                else:
                    self.pk[name] = (nv.PackedConv(b, a, self.taps, self.T, self.dev),
                                     nv.PackedConv(b, a, self.taps, self.T, self.dev, dgrad=True))
        for l in range(self.levels - 2, -1, -1):
            n = self.ch[l + 1] * self.ch[l] * self.npos
            self.pk[f'dec{l}.up'] = (torch.empty(n, dtype=self.T, device=self.dev),
                                     torch.empty(n, dtype=self.T, device=self.dev))

    def repack(self):
        """fp32 master weights -> MFMA fragment order (forward and data-gradient operators): one launch over a
        descriptor table built once (the flat master tensor and the packed buffers never move)."""
        if getattr(self, '_pack_table', None) is None:
            descs = []
            for prefix in self.stage_names():
                ci, co, _ = self.stage_io(prefix)
                for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                    name = f'{prefix}.conv{j}'
                    w = self.p(name + '.weight')
                    fwd, dg = self.pk[name]
                    if name == 'enc0.conv1':
                        descs.append(nv.make_desc(w, fwd, b, a, self.taps, 2, self.T))
                    else:
                        descs += fwd.descs(w) + dg.descs(w)
            for l in range(self.levels - 2, -1, -1):
                w = self.p(f'dec{l}.up.weight')
                fwd, dg = self.pk[f'dec{l}.up']
                descs.append(nv.make_desc(w, fwd, self.ch[l], self.ch[l + 1], self.npos, 3, self.T))
                descs.append(nv.make_desc(w, dg, self.ch[l], self.ch[l + 1], self.npos, 4, self.T))
            self._pack_table = nv.PackTable(descs, self.dev, sources=[self.flat])
        self._pack_table.run()

    # ------------------------------------------------------------------ workspace
    def workspace(self, N, D, H, W):
        key = (N, D, H, W)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        f = 2 ** (self.levels - 1)
        if H % f or W % f or (self.dim == 3 and D % f) or (self.dim == 2 and D != 1):
            raise ValueError(f'spatial size {(D, H, W)} must be divisible by {f}')
        L, ch = self.levels, self.ch
        dims = [((D >> l) if self.dim == 3 else 1, H >> l, W >> l) for l in range(L)]
        act = lambda c, v: torch.empty(N * c * v, dtype=self.T, device=self.dev)
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=self.dev)
        lib = nv.lib()
        ws = {'dims': dims}
        max_stats, max_wslab, max_bn = 0, 0, 0
        for prefix in self.stage_names():
            ci, co, l = self.stage_io(prefix)
            v = _vox(dims[l])
            d = dims[l]
            for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                name = f'{prefix}.conv{j}'
                ws['y.' + name] = act(b, v)
                skip = prefix.startswith('enc') and j == 2 and l < L - 1
                if not skip:
                    ws['z.' + name] = act(b, v)
                    ws['dz.' + name] = act(b, v)
                for k in ('scale', 'shift', 'mean', 'invstd'):
                    ws[f'{k}.{name}'] = f32(b * (N if self.gn else 1))
                if name == 'enc0.conv1':
                    max_stats = max(max_stats, lib.iunet_conv3_num_tiles(self.dim, N, *d) * b * 2)
                    max_wslab = max(max_wslab, lib.iunet_first_conv_wgrad_blocks(self.dim, N, *d) * b * 112)
                else:
                    max_stats = max(max_stats, max(lib.iunet_conv3_stats_parts(self.dim, N, *d, b, lay) for lay in (0, 2)) * b * 2)
                    if self.gn:          # per-sample rows of the conv epilogue
                        max_stats = max(max_stats, max(lib.iunet_conv3_sample_stats_rows(self.dt, self.dim, N, *d, a, b, lay) for lay in (2, 3)) * N * b * 2)
                    max_wslab = max(max_wslab, lib.iunet_conv3_wgrad_slab_floats(self.dim, N, *d, a, b))
                max_bn = max(max_bn, lib.iunet_bn_bwd_num_parts(N, v) * b * 2)
        for l in range(L):
            v = _vox(dims[l])
            if l < L - 1:
                ws[f'cat{l}'] = act(2 * ch[l], v)
                ws[f'dcat{l}'] = act(2 * ch[l], v)
                nb = lib.iunet_convT_wgrad_blocks(self.dim, N, *dims[l + 1], ch[l + 1], ch[l])
                max_wslab = max(max_wslab, nb * ch[l + 1] * ch[l] * self.npos)
                ws[f'bslab{l}'] = f32(nb * ch[l])
            if l > 0:
                ws[f'pin{l}'] = act(ch[l - 1], v)
                ws[f'dpin{l}'] = act(ch[l - 1], v)
        v0 = _vox(dims[0])
        ws['dy'] = act(max(ch[l] * _vox(dims[l]) for l in range(L)), 1)
        ws['stats'] = f32(max_stats)
        ws['wslab'] = f32(max_wslab)
        ws['bnslab'] = f32(max_bn)
        ws['bncoef'] = f32(3 * max(ch) * (N if self.gn else 1))
        nparts = lib.iunet_head_loss_num_parts(N, v0)
        ws['lslab'] = f32(nparts * self.ncls * 8)
        ws['hslab'] = f32(lib.iunet_head_loss_bwd_num_parts(N, v0, self.ncls, ch[0]) * self.ncls * (ch[0] + 1))
        ws['htmp'] = f32(self.ncls * (ch[0] + 1))
        ws['out4'] = f32(4)
        ws['coef'] = f32(self.ncls * 3)
        self._ws = {key: ws}
        return ws

    def _P(self, t, off_elems=0):
        return ctypes.c_void_p(t.data_ptr() + off_elems * self.es)

    # ------------------------------------------------------------------ forward
    def _stage_conv_fwd(self, ws, name, x_ptr, x_ss, ci, co, l, z_ptr, z_ss, N, x_raw=None, training=True, x_act=None,
                        pool=None):
        """conv -> raw output y + BatchNorm batch statistics -> scale / shift; z = relu(bn(y)) is written unless z_ptr
        is None (the consumer applies it in its loader waves).  x_act: name of the conv whose y is this conv's input
        with its BatchNorm + ReLU still to be applied (iunet_conv3_fwd_act)."""
        d = ws['dims'][l]
        v = _vox(d)
        s = nv.stream()
        y = ws['y.' + name]
        stats = ws['stats']
        gn_rows = 0
        if name == 'enc0.conv1':
            x, xs = x_raw
            w, _ = self.pk[name]
            nparts = nv.lib().iunet_conv3_num_tiles(self.dim, N, *d)
            if self.gn and self.gn_conv_stats:
                gn_rows = nparts // N          # one row per tile, a sample's tiles together: per-sample rows as they are
            nv.call('iunet_first_conv_fwd', self.dt, self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(xs),
                    self._P(y), co * v, nv.ptr(w), None, None if (self.gn and not gn_rows) else nv.ptr(stats), N, d[0], d[1], d[2], ci, co, 0, s)
        else:
            pk, _ = self.pk[name]
            lay, w = pk.pick(self.dim, N, *d, act=x_act is not None)
            nparts = nv.lib().iunet_conv3_stats_parts(self.dim, N, *d, co, lay)
            probe = self.probe if (self.probe is not None and self.probe['name'] == name) else None
            if probe is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.gn and x_act is None and self.gn_conv_stats:      # GroupNorm statistics from the conv's epilogue, per sample, where the launch has that form
                gn_rows = nv.lib().iunet_conv3_sample_stats_rows(self.dt, self.dim, N, d[0], d[1], d[2], ci, co, lay)
            if gn_rows > 0:
                nv.call('iunet_conv3_fwd_sample_stats', self.dt, self.dim, x_ptr, x_ss, self._P(y), co * v, nv.ptr(w), nv.ptr(stats),
                        N, d[0], d[1], d[2], ci, co, lay, s)
            elif x_act is None:
                nv.call('iunet_conv3_fwd', self.dt, self.dim, x_ptr, x_ss, self._P(y), co * v, nv.ptr(w), None,
                        None if self.gn else nv.ptr(stats), N, d[0], d[1], d[2], ci, co, 0, lay, s)
            else:
                nv.call('iunet_conv3_fwd_act', self.dt, self.dim, x_ptr, x_ss, self._P(y), co * v, nv.ptr(w), None,
                        nv.ptr(stats), nv.ptr(ws['scale.' + x_act]), nv.ptr(ws['shift.' + x_act]),
                        N, d[0], d[1], d[2], ci, co, 0, lay, s)
            if probe is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                probe['events'].append((e0, e1, N))
        bn = name.replace('conv', 'bn')
        if self.gn:
            if pool is not None:                # encoder stage: the normalise pass writes the activation and its max-pool
                p_ptr, p_ss, do = pool
                nv.call('iunet_gn_relu_pool_fwd_rows', self.dt, self.dim, self._P(y), co * v, z_ptr, z_ss, p_ptr, p_ss, nv.ptr(self.p(bn + '.weight')),
                        nv.ptr(self.p(bn + '.bias')), self.groups, BN_EPS, nv.ptr(stats if gn_rows > 0 else ws['bnslab']), gn_rows, nv.ptr(ws['scale.' + name]),
                        nv.ptr(ws['shift.' + name]), nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]), co, N, do[0], do[1], do[2], s)
            else:
                nv.call('iunet_gn_relu_fwd_rows', self.dt, self._P(y), co * v, z_ptr, z_ss, nv.ptr(self.p(bn + '.weight')),
                        nv.ptr(self.p(bn + '.bias')), self.groups, BN_EPS, nv.ptr(stats if gn_rows > 0 else ws['bnslab']), gn_rows, nv.ptr(ws['scale.' + name]),
                        nv.ptr(ws['shift.' + name]), nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]), co, N, v, s)
            return
        nv.call('iunet_bn_finalize', nv.ptr(stats), nparts, co, float(N) * v,
                nv.ptr(self.p(bn + '.weight')), nv.ptr(self.p(bn + '.bias')),
                nv.ptr(self.p(bn + '.running_mean')), nv.ptr(self.p(bn + '.running_var')), BN_MOMENTUM, BN_EPS,
                nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]), nv.ptr(ws['mean.' + name]),
                nv.ptr(ws['invstd.' + name]), s)
        if z_ptr is not None and pool is not None:       # encoder stage: activation and its max-pool in one pass
            p_ptr, p_ss, do = pool
            nv.call('iunet_bn_relu_pool_fwd', self.dt, self.dim, self._P(y), co * v, z_ptr, z_ss, p_ptr, p_ss,
                    nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]), co, N, do[0], do[1], do[2], s)
        elif z_ptr is not None:
            nv.call('iunet_bn_relu_fwd', self.dt, self._P(y), co * v, z_ptr, z_ss, nv.ptr(ws['scale.' + name]),
                    nv.ptr(ws['shift.' + name]), co, N, v, s)

    def _conv2_input(self, ws, stage, l, N):
        """(x_ptr, x_act, z1_ptr) of a stage's second conv.  Where that conv runs on layout 2, conv1's BatchNorm + ReLU output is never written --
        conv2 and its weight gradient read conv1's raw output and apply scale / shift / ReLU in their loader waves
        (one tensor write and one read less per stage, and no bn_relu_fwd launch)."""
        if self.fuse_act and (self.dim == 3 or self.ch[l] <= 64):
            return self._P(ws[f'y.{stage}.conv1']), f'{stage}.conv1', None
        z1 = ws[f'z.{stage}.conv1']
        return self._P(z1), None, self._P(z1)

    def forward_train(self, x, x_strides, N, D, H, W):
        ws = self.workspace(N, D, H, W)
        L, ch, dims = self.levels, self.ch, ws['dims']
        s = nv.stream()
        for l in range(L):
            v = _vox(dims[l])
            ci = self.cin if l == 0 else ch[l - 1]
            x2, act, z1p = self._conv2_input(ws, f'enc{l}', l, N)
            if l == 0:
                self._stage_conv_fwd(ws, 'enc0.conv1', None, 0, ci, ch[0], 0, z1p, ch[0] * v, N, x_raw=(x, x_strides))
            else:
                self._stage_conv_fwd(ws, f'enc{l}.conv1', self._P(ws[f'pin{l}']), ci * v, ci, ch[l], l, z1p, ch[l] * v, N)
            if l < L - 1:
                do = dims[l + 1]
                self._stage_conv_fwd(ws, f'enc{l}.conv2', x2, ch[l] * v, ch[l], ch[l], l,
                                     self._P(ws[f'cat{l}']), 2 * ch[l] * v, N, x_act=act,
                                     pool=(self._P(ws[f'pin{l + 1}']), ch[l] * _vox(do), do))
            else:
                self._stage_conv_fwd(ws, f'enc{l}.conv2', x2, ch[l] * v, ch[l], ch[l], l,
                                     self._P(ws[f'z.enc{l}.conv2']), ch[l] * v, N, x_act=act)
        for l in range(L - 2, -1, -1):
            v, vi, di = _vox(dims[l]), _vox(dims[l + 1]), dims[l + 1]
            src = ws[f'z.enc{l + 1}.conv2'] if l == L - 2 else ws[f'z.dec{l + 1}.conv2']
            wf, _ = self.pk[f'dec{l}.up']
            nv.call('iunet_convT_fwd', self.dt, self.dim, self._P(src), ch[l + 1] * vi, self._P(ws[f'cat{l}'], ch[l] * v),
                    2 * ch[l] * v, nv.ptr(wf), nv.ptr(self.p(f'dec{l}.up.bias')), N, di[0], di[1], di[2],
                    ch[l + 1], ch[l], s)
            x2, act, z1p = self._conv2_input(ws, f'dec{l}', l, N)
            self._stage_conv_fwd(ws, f'dec{l}.conv1', self._P(ws[f'cat{l}']), 2 * ch[l] * v, 2 * ch[l], ch[l], l,
                                 z1p, ch[l] * v, N)
            # the last stage's activation is read by the head only: with head_act the head kernels apply its BatchNorm + ReLU
            # while loading the raw conv output (iunet_head_loss_fwd_act / _bwd_act) and the tensor is never written
            z2 = None if (l == 0 and (self.head_act or self.gn_head)) else self._P(ws[f'z.dec{l}.conv2'])
            self._stage_conv_fwd(ws, f'dec{l}.conv2', x2, ch[l] * v, ch[l], ch[l], l, z2, ch[l] * v, N, x_act=act)
        return ws

    def loss_forward(self, ws, feat, y, w, N, vox, act=None):
        """head + softmax + loss sums + loss/metrics/coefs (device scalars in ws['out4']).  act: name of the layer whose raw
        output `feat` is (its BatchNorm + ReLU is then applied by the head kernel while loading)."""
        hw = self.p('head.weight')
        tdt = {torch.float32: 0, torch.float16: 1}[y.dtype]
        if w is not None and w.dtype != y.dtype:
            w = w.to(y.dtype)
        if act is None:
            nv.call('iunet_head_loss_fwd', self.dt, self._P(feat), self.ch[0] * vox, self.ch[0], nv.ptr(hw),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, self.kind, nv.ptr(ws['lslab']),
                    nv.ptr(ws['out4']), nv.ptr(ws['coef']), N, vox, nv.stream())
        else:      # (GroupNorm: scale / shift are per-sample rows)
            nv.call('iunet_head_loss_fwd_act_ps', self.dt, self._P(feat), self.ch[0] * vox, self.ch[0], nv.ptr(hw),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, self.kind, nv.ptr(ws['lslab']),
                    nv.ptr(ws['out4']), nv.ptr(ws['coef']), nv.ptr(ws['scale.' + act]), nv.ptr(ws['shift.' + act]),
                    int(self.gn), N, vox, nv.stream())
        return tdt, w

    # ------------------------------------------------------------------ backward
    def _stage_conv_bwd(self, ws, name, dz_ptr, dz_ss, z_ptr, z_ss, x_ptr, x_ss, ci, co, l, dx_ptr, dx_ss, N,
                        x_raw=None, x_act=None, pool_bwd=None, feeds=None, dy_ready=False):
        """Backward of one stage conv: BatchNorm + ReLU backward, weight gradient, data gradient.  `feeds`: name of the layer
        whose activation is this conv's only input (a stage's conv1 for its conv2) -- the data-gradient launch then also
        accumulates that layer's BatchNorm-backward sums in its epilogue (iunet_conv3_dgrad_bnstats), and that layer's own
        call skips its reduction pass (one read of dz and y less per conv1)."""
        d = ws['dims'][l]
        v = _vox(d)
        s = nv.stream()
        bn = name.replace('conv', 'bn')
        dy = ws['dy']
        first = name == 'enc0.conv1'
        if dy_ready:
            pass          # iunet_head_bn_bwd has written dy, dgamma and dbeta of this conv
        elif self.gn and pool_bwd is not None:
            # encoder stage: dz = skip gradient + max-pool backward of dpool, formed on the fly in both passes (never written)
            dp_ptr, dp_ss, do = pool_bwd
            nv.call('iunet_gn_relu_pool_bwd', self.dt, self.dim, dz_ptr, dz_ss, dp_ptr, dp_ss, self._P(ws['y.' + name]), co * v, self._P(dy), co * v,
                    nv.ptr(self.p(bn + '.weight')), self.groups, nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]), nv.ptr(self.g(bn + '.weight')),
                    nv.ptr(self.g(bn + '.bias')), nv.ptr(ws['bnslab']), nv.ptr(ws['bncoef']), co, N, do[0], do[1], do[2], s)
        elif self.gn:
            rows = self._bw_ready.pop(name, 0)          # > 0: the data-gradient launch that produced dz left this layer's per-sample sums in ws['stats']
            nv.call('iunet_gn_relu_bwd_rows', self.dt, dz_ptr, dz_ss, self._P(ws['y.' + name]), co * v, self._P(dy), co * v,
                    nv.ptr(self.p(bn + '.weight')), self.groups, nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]), nv.ptr(self.g(bn + '.weight')),
                    nv.ptr(self.g(bn + '.bias')), nv.ptr(ws['stats'] if rows > 0 else ws['bnslab']), rows, nv.ptr(ws['bncoef']), co, N, v, s)
        elif pool_bwd is not None:
            # encoder stage: dz = skip gradient (dz_ptr) + max-pool backward of dpool, formed on the fly in both passes
            dp_ptr, dp_ss, do = pool_bwd
            nv.call('iunet_bn_relu_pool_bwd', self.dt, self.dim, dz_ptr, dz_ss, dp_ptr, dp_ss, self._P(ws['y.' + name]), co * v,
                    self._P(dy), co * v, nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]),
                    nv.ptr(self.p(bn + '.weight')), nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(self.g(bn + '.weight')), nv.ptr(self.g(bn + '.bias')),
                    nv.ptr(ws['bnslab']), nv.ptr(ws['bncoef']), co, N, do[0], do[1], do[2], s)
        elif name in self._bw_ready:
            # pass 1 was done by the data-gradient launch that produced dz: finalize its rows, then pass 2
            nparts = self._bw_ready.pop(name)
            nv.call('iunet_bn_relu_bwd_apply', self.dt, dz_ptr, dz_ss, self._P(ws['y.' + name]), co * v,
                    None if first else self._P(dy), co * v, nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]),
                    nv.ptr(self.p(bn + '.weight')), nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(self.g(bn + '.weight')), nv.ptr(self.g(bn + '.bias')), nv.ptr(ws['stats']), nparts,
                    nv.ptr(ws['bncoef']), co, N, v, s)
        else:
            # z is not passed: the ReLU mask is recomputed from y (saves one tensor read in each of the two passes).
            # First layer: only the sums (dy = NULL) -- its single consumer, the weight gradient, applies pass 2 itself.
            nv.call('iunet_bn_relu_bwd', self.dt, dz_ptr, dz_ss, None, z_ss, self._P(ws['y.' + name]), co * v,
                    None if first else self._P(dy), co * v, nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]),
                    nv.ptr(self.p(bn + '.weight')), nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(self.g(bn + '.weight')), nv.ptr(self.g(bn + '.bias')),
                    nv.ptr(ws['bnslab']), nv.ptr(ws['bncoef']), co, N, v, s)
        gw = self.g(name + '.weight')
        if first and self.gn:
            x, xs = x_raw
            nv.call('iunet_first_conv_wgrad', self.dt, self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(xs),
                    self._P(dy), co * v, nv.ptr(ws['wslab']), nv.ptr(gw), N, d[0], d[1], d[2], ci, co, s)
        elif first:
            x, xs = x_raw
            nv.call('iunet_first_conv_wgrad_bn', self.dt, self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(xs),
                    dz_ptr, dz_ss, self._P(ws['y.' + name]), co * v, nv.ptr(ws['mean.' + name]), nv.ptr(ws['invstd.' + name]),
                    nv.ptr(ws['bncoef']), nv.ptr(ws['scale.' + name]), nv.ptr(ws['shift.' + name]),
                    nv.ptr(ws['wslab']), nv.ptr(gw), N, d[0], d[1], d[2], ci, co, s)
        else:
            if x_act is None:
                nv.call('iunet_conv3_wgrad', self.dt, self.dim, x_ptr, x_ss, self._P(dy), co * v, nv.ptr(ws['wslab']),
                        nv.ptr(gw), 1.0, N, d[0], d[1], d[2], ci, co, s)
            else:
                nv.call('iunet_conv3_wgrad_act', self.dt, self.dim, x_ptr, x_ss, self._P(dy), co * v, nv.ptr(ws['wslab']),
                        nv.ptr(gw), 1.0, nv.ptr(ws['scale.' + x_act]), nv.ptr(ws['shift.' + x_act]),
                        N, d[0], d[1], d[2], ci, co, s)
            _, pkd = self.pk[name]
            lay, wd = pkd.pick(self.dim, N, *d, bw=feeds is not None and (self.fuse_bw or self.gn_bw))      # (GroupNorm: the per-sample form of the fused sums)
            gn_rows = 0
            if self.gn and feeds is not None and self.gn_bw and (lay == 2 or (lay == 3 and self.dim == 2 and co <= 64)):
                gn_rows = nv.lib().iunet_conv3_sample_stats_rows(self.dt, self.dim, N, d[0], d[1], d[2], co, ci, lay)
            # (pick keeps the request for the fused sums only where the launch has them: layout 2, or the compact operator in 2-D up to 64 channels)
            if feeds is not None and self.fuse_bw and (lay == 2 or (lay == 3 and nv.lib().iunet_conv3_compact_ok(self.dim, N, d[0], d[1], d[2], co, ci, 0, 1))):
                nv.call('iunet_conv3_dgrad_bnstats_lay', self.dt, self.dim, self._P(dy), co * v, dx_ptr, dx_ss, nv.ptr(wd),
                        nv.ptr(ws['stats']), self._P(ws['y.' + feeds]), ci * v, nv.ptr(ws['mean.' + feeds]),
                        nv.ptr(ws['invstd.' + feeds]), nv.ptr(ws['scale.' + feeds]), nv.ptr(ws['shift.' + feeds]),
                        N, d[0], d[1], d[2], co, ci, lay, s)
                self._bw_ready[feeds] = nv.lib().iunet_conv3_stats_parts(self.dim, N, d[0], d[1], d[2], ci, 2)
            elif gn_rows > 0:
                nv.call('iunet_conv3_dgrad_sample_bnstats', self.dt, self.dim, self._P(dy), co * v, dx_ptr, dx_ss, nv.ptr(wd),
                        nv.ptr(ws['stats']), self._P(ws['y.' + feeds]), ci * v, nv.ptr(ws['mean.' + feeds]),
                        nv.ptr(ws['invstd.' + feeds]), nv.ptr(ws['scale.' + feeds]), nv.ptr(ws['shift.' + feeds]),
                        N, d[0], d[1], d[2], co, ci, lay, s)
                self._bw_ready[feeds] = gn_rows
            else:
                nv.call('iunet_conv3_fwd', self.dt, self.dim, self._P(dy), co * v, dx_ptr, dx_ss, nv.ptr(wd), None, None,
                        N, d[0], d[1], d[2], co, ci, 0, lay, s)

    def backward(self, ws, x, x_strides, y, w, tdt, N):
        L, ch, dims = self.levels, self.ch, ws['dims']
        s = nv.stream()
        v0 = _vox(dims[0])
        dfeat = ws['dz.dec0.conv2']
        nparts = nv.lib().iunet_head_loss_bwd_num_parts(N, v0, self.ncls, ch[0])
        dy_ready = None
        if self.gn_head:
            bn = 'dec0.bn2'
            nv.call('iunet_head_gn_bwd', self.dt, self._P(ws['y.dec0.conv2']), ch[0] * v0, ch[0], nv.ptr(self.p('head.weight')),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, nv.ptr(ws['coef']), 0.0, nv.ptr(self.state),
                    nv.ptr(ws['scale.dec0.conv2']), nv.ptr(ws['shift.dec0.conv2']), nv.ptr(ws['mean.dec0.conv2']), nv.ptr(ws['invstd.dec0.conv2']),
                    nv.ptr(self.p(bn + '.weight')), self.groups, nv.ptr(self.g(bn + '.weight')), nv.ptr(self.g(bn + '.bias')), self._P(ws['dy']), ch[0] * v0,
                    nv.ptr(ws['hslab']), nv.ptr(ws['bnslab']), nv.ptr(ws['bncoef']), self._P(dfeat), N, v0, s)
            dy_ready = 'dec0.conv2'
        elif self.head_bn and nv.lib().iunet_head_bn_bwd_ok(ch[0], self.ncls):
            # head backward + BatchNorm backward of dec0.conv2 in two passes over its raw output: the head's input gradient is never written
            bn = 'dec0.bn2'
            nv.call('iunet_head_bn_bwd', self.dt, self._P(ws['y.dec0.conv2']), ch[0] * v0, ch[0], nv.ptr(self.p('head.weight')),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, nv.ptr(ws['coef']), 0.0, nv.ptr(self.state),
                    nv.ptr(ws['scale.dec0.conv2']), nv.ptr(ws['shift.dec0.conv2']), nv.ptr(ws['mean.dec0.conv2']), nv.ptr(ws['invstd.dec0.conv2']),
                    nv.ptr(self.p(bn + '.weight')), nv.ptr(self.g(bn + '.weight')), nv.ptr(self.g(bn + '.bias')), self._P(ws['dy']), ch[0] * v0,
                    nv.ptr(ws['hslab']), nv.ptr(ws['bnslab']), nv.ptr(ws['bncoef']), self._P(dfeat), N, v0, s)      # (dfeat: unused by this path, its scratch)
            dy_ready = 'dec0.conv2'
        elif self.head_act:       # (the loss scale is read from the device state)
            nv.call('iunet_head_loss_bwd_dev', self.dt, self._P(ws['y.dec0.conv2']), ch[0] * v0, ch[0], nv.ptr(self.p('head.weight')),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, nv.ptr(ws['coef']),
                    nv.ptr(self.state), self._P(dfeat), ch[0] * v0, nv.ptr(ws['hslab']), nv.ptr(ws['scale.dec0.conv2']),
                    nv.ptr(ws['shift.dec0.conv2']), N, v0, s)
        else:
            nv.call('iunet_head_loss_bwd_dev', self.dt, self._P(ws['z.dec0.conv2']), ch[0] * v0, ch[0], nv.ptr(self.p('head.weight')),
                    nv.ptr(self.p('head.bias')), self.ncls, nv.ptr(y), nv.ptr(w), tdt, nv.ptr(ws['coef']),
                    nv.ptr(self.state), self._P(dfeat), ch[0] * v0, nv.ptr(ws['hslab']), None, None, N, v0, s)
        nv.call('iunet_reduce_slab', nv.ptr(ws['hslab']), nparts, self.ncls * (ch[0] + 1), nv.ptr(ws['htmp']), 1.0, 0, s)
        # slab layout: [planes][ncls][8] weight partials, then [ncls] bias partials
        nv.call('iunet_head_grad_scatter', nv.ptr(ws['htmp']), nv.ptr(self.g('head.weight')), nv.ptr(self.g('head.bias')), self.ncls, ch[0], s)
        # decoder, level 0 upwards
        for l in range(0, L - 1):
            v, vi, di = _vox(dims[l]), _vox(dims[l + 1]), dims[l + 1]
            z1, z2 = ws[f'z.dec{l}.conv1'], ws[f'z.dec{l}.conv2']
            dz1, dz2 = ws[f'dz.dec{l}.conv1'], ws[f'dz.dec{l}.conv2']
            x2, act, _ = self._conv2_input(ws, f'dec{l}', l, N)
            self._stage_conv_bwd(ws, f'dec{l}.conv2', self._P(dz2), ch[l] * v, self._P(z2), ch[l] * v, x2,
                                 ch[l] * v, ch[l], ch[l], l, self._P(dz1), ch[l] * v, N, x_act=act, feeds=f'dec{l}.conv1',
                                 dy_ready=dy_ready == f'dec{l}.conv2')
            self._stage_conv_bwd(ws, f'dec{l}.conv1', self._P(dz1), ch[l] * v, self._P(z1), ch[l] * v,
                                 self._P(ws[f'cat{l}']), 2 * ch[l] * v, 2 * ch[l], ch[l], l, self._P(ws[f'dcat{l}']),
                                 2 * ch[l] * v, N)
            # transposed conv: weight / bias gradient and data gradient
            src_name = f'enc{l + 1}.conv2' if l == L - 2 else f'dec{l + 1}.conv2'
            src, dsrc = ws['z.' + src_name], ws['dz.' + src_name]
            dup = self._P(ws[f'dcat{l}'], ch[l] * v)
            nv.call('iunet_convT_wgrad', self.dt, self.dim, self._P(src), ch[l + 1] * vi, dup, 2 * ch[l] * v,
                    nv.ptr(ws['wslab']), nv.ptr(ws[f'bslab{l}']), nv.ptr(self.g(f'dec{l}.up.weight')),
                    nv.ptr(self.g(f'dec{l}.up.bias')), N, di[0], di[1], di[2], ch[l + 1], ch[l], s)
            _, wd = self.pk[f'dec{l}.up']
            nv.call('iunet_convT_dgrad', self.dt, self.dim, dup, 2 * ch[l] * v, self._P(dsrc), ch[l + 1] * vi,
                    nv.ptr(wd), N, di[0], di[1], di[2], ch[l + 1], ch[l], s)
        # data parallel: the decoder + head gradients (the tail of the flat tensor) are complete -- their all-reduce runs
        # on RCCL's stream while the encoder backward below still computes
        if self.pg is not None:
            self.buckets.start_tail()
        # encoder, bottom level upwards
        for l in range(L - 1, -1, -1):
            v = _vox(dims[l])
            z1, dz1 = ws[f'z.enc{l}.conv1'], ws[f'dz.enc{l}.conv1']
            pool_bwd = None
            if l == L - 1:
                dz2_ptr, dz2_ss = self._P(ws[f'dz.enc{l}.conv2']), ch[l] * v
                z2_ptr, z2_ss = self._P(ws[f'z.enc{l}.conv2']), ch[l] * v
            else:
                do = dims[l + 1]
                pool_bwd = (self._P(ws[f'dpin{l + 1}']), ch[l] * _vox(do), do)      # folded into the BatchNorm backward below
                dz2_ptr, dz2_ss = self._P(ws[f'dcat{l}']), 2 * ch[l] * v
                z2_ptr, z2_ss = self._P(ws[f'cat{l}']), 2 * ch[l] * v
            x2, act, _ = self._conv2_input(ws, f'enc{l}', l, N)
            self._stage_conv_bwd(ws, f'enc{l}.conv2', dz2_ptr, dz2_ss, z2_ptr, z2_ss, x2, ch[l] * v, ch[l],
                                 ch[l], l, self._P(dz1), ch[l] * v, N, x_act=act, pool_bwd=pool_bwd, feeds=f'enc{l}.conv1')
            if l == 0:
                self._stage_conv_bwd(ws, 'enc0.conv1', self._P(dz1), ch[0] * v, self._P(z1), ch[0] * v, None, 0,
                                     self.cin, ch[0], 0, None, 0, N, x_raw=(x, x_strides))
            else:
                self._stage_conv_bwd(ws, f'enc{l}.conv1', self._P(dz1), ch[l] * v, self._P(z1), ch[l] * v,
                                     self._P(ws[f'pin{l}']), ch[l - 1] * v, ch[l - 1], ch[l], l,
                                     self._P(ws[f'dpin{l}']), ch[l - 1] * v, N)
            if l == L - 1 and L > 1 and self.pg is not None and self._bottom_start is not None:
                self.buckets.start(self._bottom_start, self._dec_start)

    # ------------------------------------------------------------------ optimiser
    def optimizer_step(self):
        s = nv.stream()
        n = self.flat.numel()
        world = self.buckets.finish() if self.pg is not None else 1
        # overflow check (fp16), this step's coefficients, AdamW (skipped on overflow), GradScaler's back-off / growth and the step
        # count: all on the device state -- nothing is read back
        nv.call('iunet_adamw_step_dev', nv.ptr(self.flat), nv.ptr(self.grad), nv.ptr(self.m), nv.ptr(self.v), n,
                float(self.lr), self.betas[0], self.betas[1], self.eps, self.wd, nv.ptr(self.state),
                int(self.T == torch.float16), float(world), s)
        self.repack()
        self.model._packed_sig = None          # weights changed behind torch's version counters

    # ------------------------------------------------------------------ public steps
    def _prep(self, X, y, w):
        X = X.to(self.dev).contiguous()
        y = y.to(self.dev).contiguous()
        w = None if w is None else w.to(self.dev).contiguous()
        if y.dtype not in (torch.float16, torch.float32):
            y = y.float()
        N = X.shape[0]
        sp = tuple(X.shape[2:])
        D, H, W = sp if self.dim == 3 else (1,) + sp
        vox = D * H * W
        if X.dtype not in nv.IN_DTYPE_CODE:
            X = X.float()
        return X, y, w, N, D, H, W, vox, (self.cin * vox, vox, H * W, W, 1)

    def train_step(self, X, y, w=None, sync=True):
        """One optimisation step (unet.py:88-102 + backward + AdamW).  X [N,C,*sp], y / w
        [N,ncls,*sp] (fp16 or fp32, the loader's contract loader.py:142-154)."""
        self.sync_weights()
        X, y, w, N, D, H, W, vox, xs = self._prep(X, y, w)
        h = self._handle()
        if h is not None and self.pg is not None:
            # data parallel, C-sequenced: forward + backward as one call whose host callback starts the gradient buckets' all-reduce
            # between the backward's launches (decoder + head first, then the bottom encoder level: dp.GradBuckets), the rest of the
            # vector behind the call, then the update with the summed gradient
            out4 = h.forward_backward(X, xs, y, w, N, D, H, W, buckets=self.buckets, bottom=(self._bottom_start, self._dec_start))
            world = self.buckets.finish()
            h.update(float(self.lr), self.betas, self.eps, self.wd, float(world))
            self._py_stale = True
            self.model._packed_sig = None
        elif h is not None:
            # the whole step as ONE C call (csrc/train_net.hip: the same launches in the same order, sequenced in C++)
            out4 = h.step(X, xs, y, w, N, D, H, W, float(self.lr), self.betas, self.eps, self.wd)
            self._py_stale = True                  # the Python sequence's packed operators are behind the weights now
            self.model._packed_sig = None
        else:
            self._refresh()
            ws = self.forward_train(X, xs, N, D, H, W)
            if self.head_act or self.gn_head:
                tdt, w = self.loss_forward(ws, ws['y.dec0.conv2'], y, w, N, vox, act='dec0.conv2')
            else:
                tdt, w = self.loss_forward(ws, ws['z.dec0.conv2'], y, w, N, vox)
            self.backward(ws, X, xs, y, w, tdt, N)
            self.optimizer_step()
            out4 = ws['out4']
        if sync:
            o = out4.tolist()
            return {'Loss': o[0], 'Dice': o[1], 'IoU': o[2], 'MCC': o[3]}
        return out4

    def _handle(self):
        """The C++-sequenced step (TrainHandle over iunet_train_*), or None where it does not apply: a timing probe attached,
        IUNET_PY_TRAIN=1 (A/B switch), and the FIRST step (a model that trains one step -- a smoke test -- never pays for the handle's own
        copy of the packed operators).  Data parallel runs through it too (iunet_train_forward_backward_hooks: the gradient buckets start
        their all-reduce from a host callback between the backward's launches); IUNET_PY_DP=1 keeps that path on the Python sequence."""
        self._steps_seen = getattr(self, '_steps_seen', 0) + 1
        if (self.pg is not None and os.environ.get('IUNET_PY_DP')) or self.probe is not None or os.environ.get('IUNET_PY_TRAIN') \
                or not self.use_handle or self._steps_seen < 2:
            return None
        if getattr(self, '_h', None) is None:
            self._h = TrainHandle(self)
        return self._h

    def _refresh(self):
        """Before a Python-sequenced forward: re-pack this sequence's operators if C++-sequenced steps moved the weights since."""
        if getattr(self, '_py_stale', False):
            self.repack()
            self._py_stale = False

    def _eval_engine(self):
        """The folded-BatchNorm forward in the TRAINING dtype (its features feed the fused head + loss kernel): the module's own
        engine when it predicts in that dtype, else (the default module predicts in split precision) one kept here."""
        m = self.model
        if m.infer_dtype == self.T:
            return m.engine('eval')
        if getattr(self, '_eval_eng', None) is None:
            from .engine import Engine
            self._eval_eng = Engine(self.dim, self.levels, m.base, self.cin, self.ncls, self.T, self.dev, norm=self.norm,
                                    groups=self.groups)
        sig = (m._signature(), getattr(self, '_steps_seen', 0))
        if sig != getattr(self, '_eval_sig', None):              # re-pack only when a step (or anyone else) moved the weights: a validation
            self._eval_eng.load_eval(m.named_tensors())          # pass over many batches folds once (one launch over a cached table) and
            self._eval_sig = sig                                 # runs from its second batch as one C call per batch (net_graph)
        return self._eval_eng

    def sync_weights(self):
        """Re-pack the operators when something other than optimizer_step changed the parameters (an external optimiser stepping
        the module's parameters -- UNet.configure_optimizers -- or load_state_dict): torch's version counters tell."""
        ver = sum(self.p(n)._version for n in self.names)
        if ver != getattr(self, '_seen_version', None):
            if getattr(self, '_seen_version', None) is not None:
                self.repack()
                if getattr(self, '_h', None) is not None:
                    self._h.repack()
                self.model._packed_sig = None
            self._seen_version = ver

    def step_forward(self, X, y, w=None):
        """Forward half of a training step (unet.py:88-102): -> (out4 = [Loss, Dice, IoU, MCC] device tensor, state for
        step_backward).  UNet.training_step wraps the pair in an autograd function."""
        self.sync_weights()
        self._refresh()
        X, y, w, N, D, H, W, vox, xs = self._prep(X, y, w)
        ws = self.forward_train(X, xs, N, D, H, W)
        if self.head_act or self.gn_head:
            tdt, w = self.loss_forward(ws, ws['y.dec0.conv2'], y, w, N, vox, act='dec0.conv2')
        else:
            tdt, w = self.loss_forward(ws, ws['z.dec0.conv2'], y, w, N, vox)
        return ws['out4'], (ws, X, xs, y, w, tdt, N)

    def step_backward(self, state):
        """Backward half: the flat fp32 gradient of loss_scale x loss in self.grad (all-reduced over the process group, if any).
        -> (flat gradient of the loss itself as a new tensor, finite?).  fp16: a non-finite gradient halves the loss scale
        (GradScaler's back-off) and comes back as zeros; 2000 finite steps in a row double it (GradScaler's growth).  The caller owns the
        optimiser here (UNet.configure_optimizers): `last_backward_ok` tells it whether to step -- torch's AdamW applied to an all-zero
        gradient still decays the weights and moves its moments, so a loop that wants GradScaler's semantics skips
        `optimizer.step()` when it is False."""
        ws, X, xs, y, w, tdt, N = state
        self.backward(ws, X, xs, y, w, tdt, N)
        world = self.buckets.finish() if self.pg is not None else 1
        # the scale lives on the device: bf16 trains at a fixed 1.0 and must not pay a host read per step for it
        scale = self.loss_scale if self.T == torch.float16 else getattr(self, '_fixed_scale', None)
        if scale is None:
            scale = self._fixed_scale = self.loss_scale
        g = self.grad * (1.0 / (scale * world))
        ok = True
        if self.T == torch.float16:
            ok = bool(torch.isfinite(g).all().item())
            if not ok:
                if self.dynamic_scale:
                    self.loss_scale = max(scale * 0.5, 1.0)
                    self._good_backwards = 0
                g.zero_()
            elif self.dynamic_scale:
                self._good_backwards = getattr(self, '_good_backwards', 0) + 1
                if self._good_backwards >= 2000:
                    self.loss_scale = scale * 2.0
                    self._good_backwards = 0
        self.last_backward_ok = ok
        return g, ok

    def eval_step(self, X, y, w=None, sync=True):
        """validation_step (unet.py:104-116): eval-mode BatchNorm (running statistics)."""
        self.sync_weights()
        X, y, w, N, D, H, W, vox, xs = self._prep(X, y, w)
        eng = self._eval_engine()
        g = eng._graph() if hasattr(eng, '_graph') else None
        if g is not None and eng.probe is None and not os.environ.get('IUNET_PY_EVAL'):
            # the validation step as ONE C call (iunet_net_eval_step: the handle's eval-mode forward + the fused head / loss kernel); from
            # the second batch on the same weights, as every use of the prediction handle
            tdt = {torch.float32: 0, torch.float16: 1}[y.dtype]
            if w is not None and w.dtype != y.dtype:
                w = w.to(y.dtype)
            out4 = g.eval_step(X, xs, N, D, H, W, y, w, tdt, self.kind)
            if not sync:
                return out4
            o = out4.tolist()
            return {'Loss': o[0], 'Dice': o[1], 'IoU': o[2], 'MCC': o[3]}
        feat = eng.infer(X, xs, N, D, H, W, features_only=True)
        ws = self.workspace(N, D, H, W)
        self.loss_forward(ws, feat, y, w, N, vox)
        if not sync:
            return ws['out4']                 # device tensor [loss, dice, iou, mcc], overwritten by the next step: clone to keep
        o = ws['out4'].tolist()
        return {'Loss': o[0], 'Dice': o[1], 'IoU': o[2], 'MCC': o[3]}


_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)       # iunet_train_hook


class TrainHandle:
    """iunet_train_* handle (csrc/train_net.hip) over a TrainEngine's own device vectors: the flat parameters / gradient / AdamW moments,
    the module's BatchNorm running statistics and the device training state are SHARED with the Python-sequenced path (either may run the
    next step); the packed operators and the step workspace are the handle's own."""

    def __init__(self, te):
        self.te = te
        self.lib = nv.lib()
        self.h = ctypes.c_void_p()
        m = te.model
        nv.call('iunet_train_create_ex', te.dim, te.levels, m.base, te.cin, te.ncls, te.dt, te.kind, 1 if te.gn else 0, int(te.groups), ctypes.byref(self.h))
        n = self.lib.iunet_train_num_params(self.h)
        if n != te.flat.numel():
            raise RuntimeError(f'iunet_train: {n} parameters, the engine holds {te.flat.numel()}')
        for i in range(self.lib.iunet_train_num_tensors(self.h)):        # the two flat layouts are the same order
            name = ctypes.create_string_buffer(96)
            off, cnt = ctypes.c_longlong(), ctypes.c_longlong()
            nv.call('iunet_train_param', self.h, i, name, 96, ctypes.byref(off), ctypes.byref(cnt))
            if te.offsets[name.value.decode()] != (off.value, cnt.value):
                raise RuntimeError(f'iunet_train: flat layout mismatch at {name.value.decode()}')
        bns = []
        for prefix in te.stage_names():
            for j in (1, 2):
                bns += [te.p(f'{prefix}.bn{j}.running_mean'), te.p(f'{prefix}.bn{j}.running_var')]
        assert len(bns) == 2 * self.lib.iunet_train_num_bn(self.h)
        self._running = bns
        arr = (ctypes.c_void_p * len(bns))(*[t.data_ptr() for t in bns])
        self.packed = torch.empty(self.lib.iunet_train_packed_bytes(self.h), dtype=torch.uint8, device=te.dev)
        nv.call('iunet_train_bind', self.h, nv.ptr(te.flat), nv.ptr(te.grad), nv.ptr(te.m), nv.ptr(te.v), arr, nv.ptr(self.packed),
                nv.ptr(te.state), nv.stream())
        self._ws = {}
        self.out4 = torch.empty(4, dtype=torch.float32, device=te.dev)

    def __del__(self):
        try:
            if self.h:
                self.lib.iunet_train_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def repack(self):
        nv.call('iunet_train_repack', self.h, nv.stream())

    def workspace(self, N, D, H, W):
        key = (N, D, H, W)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.iunet_train_workspace_bytes(self.h, N, D, H, W)
            if nbytes <= 0:
                raise ValueError(f'spatial size {(D, H, W)} must be divisible by {2 ** (self.te.levels - 1)}')
            self._ws = {key: torch.empty(nbytes, dtype=torch.uint8, device=self.te.dev)}
            ws = self._ws[key]
        return ws

    def _targets(self, y, w):
        tdt = {torch.float32: 0, torch.float16: 1}[y.dtype]
        if w is not None and w.dtype != y.dtype:
            w = w.to(y.dtype)
        return tdt, w

    def forward_backward(self, X, xs, y, w, N, D, H, W, buckets=None, bottom=None):
        """buckets (dp.GradBuckets): the C call's host callback starts the all-reduce of the decoder + head gradients (stage 0) and of the
        bottom encoder level's (stage 1, bottom = (lo, hi) of that run in the flat vector or (None, ...)) between the backward's launches."""
        tdt, w = self._targets(y, w)
        if buckets is None:
            nv.call('iunet_train_forward_backward', self.h, nv.ptr(X), nv.IN_DTYPE_CODE[X.dtype], nv.ll_array(xs), nv.ptr(y), nv.ptr(w), tdt,
                    N, D, H, W, nv.ptr(self.workspace(N, D, H, W)), nv.ptr(self.out4), nv.stream())
            return self.out4
        err = []

        def hook(_ctx, stage):
            try:                                    # (an exception must not unwind through the C frame)
                if stage == 0:
                    buckets.start_tail()
                elif bottom is not None and bottom[0] is not None:
                    buckets.start(bottom[0], bottom[1])
            except BaseException as e:              # noqa: BLE001
                err.append(e)
        cb = _HOOK(hook)
        nv.call('iunet_train_forward_backward_hooks', self.h, nv.ptr(X), nv.IN_DTYPE_CODE[X.dtype], nv.ll_array(xs), nv.ptr(y), nv.ptr(w), tdt,
                N, D, H, W, nv.ptr(self.workspace(N, D, H, W)), nv.ptr(self.out4), ctypes.cast(cb, ctypes.c_void_p), None, nv.stream())
        if err:
            raise err[0]
        return self.out4

    def update(self, lr, betas, eps, wd, world=1.0):
        nv.call('iunet_train_update', self.h, lr, betas[0], betas[1], eps, wd, world, nv.stream())

    def step(self, X, xs, y, w, N, D, H, W, lr, betas, eps, wd):
        tdt, w = self._targets(y, w)
        nv.call('iunet_train_step', self.h, nv.ptr(X), nv.IN_DTYPE_CODE[X.dtype], nv.ll_array(xs), nv.ptr(y), nv.ptr(w), tdt,
                N, D, H, W, nv.ptr(self.workspace(N, D, H, W)), lr, betas[0], betas[1], eps, wd, nv.ptr(self.out4), nv.stream())
        return self.out4
