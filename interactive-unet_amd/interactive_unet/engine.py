"""Host-side plan of the native U-Net: owns packed weights and activation workspaces
(PyTorch-ROCm tensors = device memory + streams) and sequences the libiunet kernels.

Canonical network (replaces smp.Unet behind unet.py:33-61):
levels L, channels base*2^l, stage = 2 x [conv 3^d -> BatchNorm -> ReLU], max-pool 2,
decoder = ConvTranspose k2 s2 -> concat(skip, up) -> stage, head 1x1 -> softmax.

Activations live in HBM as channel-blocked NHWC ("NHWC8c": C/8 planes of [D][H][W][8]);
the skip concat is free (encoder conv2 and the transposed conv write into the two halves
of one buffer).  External tensors stay NCHW / uint8 / strided: the first conv and the head
read / write them directly.
"""
import ctypes
import math
import os

import torch

from . import _native as nv

BN_EPS = 1e-5


def _vox(dims):
    return dims[0] * dims[1] * dims[2]


class F8Conv:
    """A stage conv's operator for the fp8 matrix cores: e4m3 bytes (K16 order) + fp32 per-output-channel scales."""

    def __init__(self, bytes_, scale):
        self.bytes, self.scale = bytes_, scale


class Engine:
    def __init__(self, dim=2, levels=4, base=32, cin=1, ncls=2, act_dtype=torch.float16, device='cuda',
                 weight_dtype=None, norm='batch', groups=8, act_quant=None):
        if dim not in (2, 3):
            raise ValueError('dim must be 2 or 3')
        if base % 32 != 0:
            raise NotImplementedError('native U-Net needs base channels to be a multiple of 32')
        if not (1 <= cin <= 4):
            raise NotImplementedError('native U-Net supports 1..4 input channels')
        if not (2 <= ncls <= 10):
            raise NotImplementedError('native U-Net supports 2..10 classes (app.py:162)')
        self.dim, self.levels, self.base, self.cin, self.ncls = dim, levels, base, cin, ncls
        self.act_dtype = act_dtype
        if weight_dtype not in (None, 'fp8_e4m3'):
            raise ValueError("weight_dtype must be None (= activation dtype) or 'fp8_e4m3'")
        self.weight_dtype = weight_dtype      # 'fp8_e4m3': inference weights on the OCP e4m3 grid (config C5)
        # act_quant (fp8 weights only).  True (default, "W8A8"): the stage convs run on the fp8 matrix cores -- gfx950 has no mixed
        # fp8 x bf16 MFMA, so their 16-bit activations are ALSO rounded to unscaled e4m3 (3 mantissa bits, saturating at +-448) on
        # the way into LDS; measured on C5: mean |dp| 1.5e-2 against fp32, class map equal on ~95 % (tests/test_gpu_f8.py).
        # False ("W8A16", BASELINE C5's literal "fp8 weights / bf16 activations"): the operators are e4m3 VALUES times a per-channel
        # power of two, stored in the activation dtype and multiplied on the 16-bit matrix cores with unquantised activations.
        if act_quant is not None and not weight_dtype:
            raise ValueError("act_quant selects between the two fp8-weight modes: give weight_dtype='fp8_e4m3'")
        self.act_quant = bool(weight_dtype) if act_quant is None else bool(act_quant)
        if norm not in ('batch', 'group'):
            raise ValueError("norm must be 'batch' or 'group'")
        if norm == 'group' and weight_dtype:
            raise NotImplementedError('the fp8 operator format folds the norm into the weights: BatchNorm only')
        if norm == 'group' and base % groups:
            raise ValueError(f'{groups} groups do not divide {base} channels')
        # 'group': GroupNorm(groups) + ReLU after every stage conv instead of (folded) BatchNorm -- per-sample statistics, so the
        # convs write their raw output and a statistics + normalise pass follows (iunet_gn_relu_fwd)
        self.norm, self.groups = norm, groups
        self.dt = nv.DTYPE_CODE[act_dtype]
        self.device = torch.device(device)
        self.ch = [base * 2 ** l for l in range(levels)]
        self.taps = 3 ** dim
        self.npos = 2 ** dim
        self.packed = None
        self._ws_cache = {}
        self._f8_ws = None
        self.probe = None          # {'name': layer, 'events': []}: timing hook of one layer's launches (bench.py roofline)
        self.use_graph = True      # False: every forward sequenced from Python (tests compare the two)
        self._g, self._gparams, self._g_dirty, self._g_fwd = None, None, False, 0      # the C++-sequenced forward (_graph)
        nv.lib()   # fail loudly now if the HIP library is missing

    def _graph(self):
        """The C++-sequenced forward (net_graph.NetGraph) on this engine's current parameters, or None where the handle level does not
        apply (GroupNorm, fp8 operators, IUNET_PY_GRAPH=1).  The handle packs its own copy of the operators (one copy of the parameters
        + ~40 launches), so it is loaded at the SECOND forward on the same parameters: a training loop that predicts once per optimiser
        step, or a validation pass that only asks for features, never pays for it; a slice / block / volume prediction does once."""
        from . import net_graph
        self._g_fwd += 1
        if self._g_fwd < 2:
            return None
        if not net_graph.ENABLED or not self.use_graph or self.norm != 'batch' or self.weight_dtype or not (2 <= self.levels <= 6) or self._gparams is None:
            return None
        if self._g is None:
            self._g = net_graph.NetGraph(self.dim, self.levels, self.base, self.cin, self.ncls, self.dt, self.device)
        if self._g_dirty:
            self._g.set_params(self._gparams)
            self._g_dirty = False
            self._ws_cache.clear()         # the handle has its own workspace: the Python sequence's buffers go back to the allocator
        return self._g

    # ------------------------------------------------------------------ weights
    def stage_names(self):
        names = [f'enc{l}' for l in range(self.levels)] + [f'dec{l}' for l in range(self.levels - 2, -1, -1)]
        return names

    def stage_io(self, prefix):
        l = int(prefix[3:])
        if prefix.startswith('enc'):
            ci = self.cin if l == 0 else self.ch[l - 1]
        else:
            ci = 2 * self.ch[l]
        return ci, self.ch[l]

    def _source(self, params, name):
        """fp32 device tensor the pack kernel reads: the parameter itself when it already lives on the device
        (stable address, in-place optimiser updates are seen), otherwise a persistent staging copy."""
        t = params[name].detach()
        if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous():
            return t
        st = self._stage.get(name)
        if st is None or st.shape != t.shape:
            st = self._stage[name] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
        st.copy_(t)
        return st

    def load_eval(self, params):
        """Fold eval-mode BatchNorm into the stage convs and pack everything into MFMA fragment order: ONE
        launch over a device-resident descriptor table (iunet_pack_batch), rebuilt only when a source tensor
        moves.  `params`: {name: fp32 tensor}."""
        if not hasattr(self, '_stage'):
            self._stage, self._eval_sig, self._eval_table = {}, None, None
        self._gparams, self._g_dirty, self._g_fwd = params, True, 0
        src = {}
        for prefix in self.stage_names():
            for j in (1, 2):
                src[f'{prefix}.conv{j}.weight'] = self._source(params, f'{prefix}.conv{j}.weight')
                for k in ('weight', 'bias', 'running_mean', 'running_var'):
                    src[f'{prefix}.bn{j}.{k}'] = self._source(params, f'{prefix}.bn{j}.{k}')
        for l in range(self.levels - 1):
            src[f'dec{l}.up.weight'] = self._source(params, f'dec{l}.up.weight')
            src[f'dec{l}.up.bias'] = self._source(params, f'dec{l}.up.bias')
        src['head.weight'] = self._source(params, 'head.weight')
        src['head.bias'] = self._source(params, 'head.bias')
        sig = tuple(t.data_ptr() for t in src.values())
        if sig != self._eval_sig:
            P, descs, keep_q = {}, [], []
            for prefix in self.stage_names():
                ci, co = self.stage_io(prefix)
                for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                    w = src[f'{prefix}.conv{j}.weight']
                    bn = [src[f'{prefix}.bn{j}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')]
                    if self.norm == 'group':                 # nothing folds: raw operator, gamma / beta go to the norm pass
                        if prefix == 'enc0' and j == 1:
                            dst = torch.empty(nv.lib().iunet_pack_first_conv_elems(b, a, self.taps), dtype=self.act_dtype,
                                              device=self.device)
                            descs.append(nv.make_desc(w, dst, b, a, self.taps, 2, self.act_dtype))
                        else:
                            dst = nv.PackedConv(b, a, self.taps, self.act_dtype, self.device)
                            descs += dst.descs(w)
                        P[f'{prefix}.conv{j}'] = (dst, None, bn[0], bn[1])
                        continue
                    bias = torch.empty(b, dtype=torch.float32, device=self.device)
                    qs = torch.empty(b, dtype=torch.float32, device=self.device) if self.weight_dtype else None
                    if prefix == 'enc0' and j == 1:
                        dst = torch.empty(nv.lib().iunet_pack_first_conv_elems(b, a, self.taps), dtype=self.act_dtype,
                                          device=self.device)
                        descs.append(nv.make_desc(w, dst, b, a, self.taps, 2, self.act_dtype, bn=bn, bias_out=bias,
                                                  eps=BN_EPS, qscale=qs))
                    elif self.weight_dtype and self.act_quant:
                        # config C5: the stage convolutions run on the fp8 matrix cores -- operator stored as e4m3 bytes +
                        # per-output-channel scales (descriptor kind 5 of the same table)
                        dst = torch.empty(nv.lib().iunet_f8_pack_conv3_bytes(b, a, self.taps), dtype=torch.uint8, device=self.device)
                        descs.append(nv.make_desc(w, dst, b, a, self.taps, 5, self.act_dtype, bn=bn, bias_out=bias, eps=BN_EPS, qscale=qs))
                        P[f'{prefix}.conv{j}'] = (F8Conv(dst, qs), bias)
                        keep_q.append(qs)
                        continue
                    else:
                        dst = nv.PackedConv(b, a, self.taps, self.act_dtype, self.device)
                        descs += dst.descs(w, bn, bias, BN_EPS, qs)
                    P[f'{prefix}.conv{j}'] = (dst, bias)
                    keep_q.append(qs)
            for l in range(self.levels - 2, -1, -1):
                w = src[f'dec{l}.up.weight']
                dst = torch.empty(w.numel(), dtype=self.act_dtype, device=self.device)
                qs = torch.empty(self.ch[l], dtype=torch.float32, device=self.device) if self.weight_dtype else None
                keep_q.append(qs)
                descs.append(nv.make_desc(w, dst, self.ch[l], self.ch[l + 1], self.npos, 3, self.act_dtype, qscale=qs))
                P[f'dec{l}.up'] = (dst, src[f'dec{l}.up.bias'])
            P['head'] = (src['head.weight'].reshape(self.ncls, self.ch[0]), src['head.bias'])
            self._eval_table = nv.PackTable(descs, self.device, sources=list(src.values()) + keep_q)
            self._eval_sig = sig
            self.packed = P
        self._eval_table.run()

    # ------------------------------------------------------------------ workspace
    def q_planes(self):
        """True when the activations between fp8 convolutions travel as e4m3 planes: config C5 in 3-D with every stage conv on the
        K = 128 operator order (IUNET_F8_Q=0: 16-bit tensors everywhere -- the A/B switch; the results are the same bits)."""
        if getattr(self, '_qp', None) is None:            # decided once per engine (its workspaces depend on it)
            self._qp = bool(self.weight_dtype and self.act_quant and self.dim == 3 and self.norm != 'group'
                            and os.environ.get('IUNET_F8_Q', '1') != '0' and nv.lib().iunet_f8_pack_order(27, self.ch[0]) == 1)
        return self._qp

    def level_dims(self, D, H, W):
        out = []
        for l in range(self.levels):
            f = 2 ** l
            out.append((D // f if self.dim == 3 else 1, H // f, W // f))
        return out

    def check_shape(self, D, H, W):
        f = 2 ** (self.levels - 1)
        if H % f or W % f or (self.dim == 3 and D % f) or (self.dim == 2 and D != 1):
            raise ValueError(f'spatial size {(D, H, W)} must be divisible by {f} (and D == 1 in 2-D)')

    def workspace(self, N, D, H, W):
        key = (N, D, H, W)
        ws = self._ws_cache.get(key)
        if ws is None:
            self.check_shape(D, H, W)
            dims = self.level_dims(D, H, W)
            mk = lambda c, v: torch.empty(N * c * v, dtype=self.act_dtype, device=self.device)
            # fp8 network on the K = 128 path: every tensor that only fp8 convolutions read is stored as e4m3 planes (one byte per
            # element: include/iunet.h, format 1); what a transposed conv or the head reads (b) stays 16-bit
            mq = (lambda c, v: torch.empty(N * c * v, dtype=torch.uint8, device=self.device)) if self.q_planes() else mk
            ws = {'dims': dims}
            for l in range(self.levels):
                v = _vox(dims[l])
                ws[f'a{l}'] = mq(self.ch[l], v)
                ws[f'b{l}'] = mk(self.ch[l], v)
                if l < self.levels - 1:
                    ws[f'cat{l}'] = mq(2 * self.ch[l], v)
                if l > 0:
                    ws[f'pin{l}'] = mq(self.ch[l - 1], v)
            if self.norm == 'group':
                f32 = lambda n: torch.empty(n, dtype=torch.float32, device=self.device)
                ws['raw'] = mk(max(self.ch[l] * _vox(dims[l]) for l in range(self.levels)), 1)
                # fused form (one sample at a time): raw outputs of a stage's two convs, the convs' statistics rows, two
                # (scale, shift, mean, invstd) sets
                one = max(self.ch[l] * _vox(dims[l]) for l in range(self.levels))
                ws['rawA'] = torch.empty(one, dtype=self.act_dtype, device=self.device)
                ws['rawB'] = torch.empty(one, dtype=self.act_dtype, device=self.device)
                parts = max(max(nv.lib().iunet_conv3_num_tiles(self.dim, 1, *dims[l]),
                                nv.lib().iunet_conv3_stats_parts(self.dim, 1, *dims[l], self.ch[l], 2)) * self.ch[l] for l in range(self.levels))
                ws['gstats'] = f32(2 * parts)
                ws['gnA'] = [f32(max(self.ch)) for _ in range(4)]
                ws['gnB'] = [f32(max(self.ch)) for _ in range(4)]
                ws['gn'] = [f32(N * max(self.ch)) for _ in range(4)]
                ws['gnslab'] = f32(max(nv.lib().iunet_gn_num_parts(N, _vox(dims[l])) * self.ch[l] * 2 for l in range(self.levels)))
            if len(self._ws_cache) > 4:
                self._ws_cache.clear()
            self._ws_cache[key] = ws
        return ws

    # ------------------------------------------------------------------ forward (inference)
    def _group_norm(self, name, ws, y_ptr, y_ss, N, dims, co, s):
        """z = relu(group_norm(raw conv output in ws['raw'])) -> y_ptr (which may be a strided view of a concat buffer)."""
        gamma, beta = self.packed[name][2], self.packed[name][3]
        sc, sh, mu, ist = ws['gn']
        v = _vox(dims)
        nv.call('iunet_gn_relu_fwd', self.dt, nv.ptr(ws['raw']), co * v, y_ptr, y_ss, nv.ptr(gamma), nv.ptr(beta), self.groups,
                BN_EPS, nv.ptr(ws['gnslab']), nv.ptr(sc), nv.ptr(sh), nv.ptr(mu), nv.ptr(ist), co, N, v, s)

    def _gn_stage(self, ws, stage, x_ptr, ci, co, d, z_ptr, s, first=None, pool=None):
        """One stage (two convs, GroupNorm + ReLU after each) of ONE sample, fused as the BatchNorm training forward is: the convs'
        epilogues deliver the per-channel sums (iunet_gn_finalize turns them into the sample's scale / shift), conv1's activation
        is applied by conv2's loader waves and never written, conv2's is written once (with the max-pool of an encoder stage in the
        same pass).  x_ptr: this sample's input planes (first: (x tensor, strides, sample) for the network's first conv);
        z_ptr: this sample's output planes; pool = (pooled output pointer, pooled dims)."""
        L = nv.lib()
        v = _vox(d)
        st = ws['gstats']
        for j, (a, b, src) in enumerate(((ci, co, x_ptr), (co, co, None)), 1):
            name = f'{stage}.conv{j}'
            pk, _, gamma, beta = self.packed[name]
            y = ws['rawA'] if j == 1 else ws['rawB']
            sc, sh, mu, ist = ws['gnA'] if j == 1 else ws['gnB']
            if first is not None and j == 1:
                x, xs, n = first
                nparts = L.iunet_conv3_num_tiles(self.dim, 1, *d)
                nv.call('iunet_first_conv_fwd', self.dt, self.dim, ctypes.c_void_p(x.data_ptr() + n * xs[0] * x.element_size()),
                        nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(xs), nv.ptr(y), b * v, nv.ptr(pk), None, nv.ptr(st),
                        1, d[0], d[1], d[2], a, b, 0, s)
            elif j == 1:
                lay, wpk = pk.pick(self.dim, 1, *d)
                nparts = L.iunet_conv3_stats_parts(self.dim, 1, *d, b, lay)
                nv.call('iunet_conv3_fwd', self.dt, self.dim, src, a * v, nv.ptr(y), b * v, nv.ptr(wpk), None, nv.ptr(st),
                        1, d[0], d[1], d[2], a, b, 0, lay, s)
            else:
                scA, shA = ws['gnA'][0], ws['gnA'][1]
                lay, wpk = pk.pick(self.dim, 1, *d, act=True)
                if lay in (2, 3):            # conv1's GroupNorm + ReLU in this conv's loader waves
                    nparts = L.iunet_conv3_stats_parts(self.dim, 1, *d, b, lay)
                    nv.call('iunet_conv3_fwd_act', self.dt, self.dim, nv.ptr(ws['rawA']), a * v, nv.ptr(y), b * v, nv.ptr(wpk), None,
                            nv.ptr(st), nv.ptr(scA), nv.ptr(shA), 1, d[0], d[1], d[2], a, b, 0, lay, s)
                else:                        # layouts without the fused input activation: materialise it (in place of the raw output)
                    nv.call('iunet_bn_relu_fwd', self.dt, nv.ptr(ws['rawA']), a * v, nv.ptr(ws['rawA']), a * v, nv.ptr(scA), nv.ptr(shA),
                            a, 1, v, s)
                    lay, wpk = pk.pick(self.dim, 1, *d)
                    nparts = L.iunet_conv3_stats_parts(self.dim, 1, *d, b, lay)
                    nv.call('iunet_conv3_fwd', self.dt, self.dim, nv.ptr(ws['rawA']), a * v, nv.ptr(y), b * v, nv.ptr(wpk), None,
                            nv.ptr(st), 1, d[0], d[1], d[2], a, b, 0, lay, s)
            nv.call('iunet_gn_finalize', nv.ptr(st), nparts, b, self.groups, v, nv.ptr(gamma), nv.ptr(beta), BN_EPS,
                    nv.ptr(sc), nv.ptr(sh), nv.ptr(mu), nv.ptr(ist), s)
        scB, shB = ws['gnB'][0], ws['gnB'][1]
        if pool is not None:
            p_ptr, do = pool
            nv.call('iunet_bn_relu_pool_fwd', self.dt, self.dim, nv.ptr(ws['rawB']), co * v, z_ptr, co * v, p_ptr, co * _vox(do),
                    nv.ptr(scB), nv.ptr(shB), co, 1, do[0], do[1], do[2], s)
        else:
            nv.call('iunet_bn_relu_fwd', self.dt, nv.ptr(ws['rawB']), co * v, z_ptr, co * v, nv.ptr(scB), nv.ptr(shB), co, 1, v, s)

    def _infer_gn_features(self, ws, x, x_strides, N, s):
        """The GroupNorm network up to the head's input (ws['b0']), one sample at a time through _gn_stage (GroupNorm statistics are
        per sample: a launch of one sample delivers them in the conv epilogue's rows)."""
        dims, L, ch = ws['dims'], self.levels, self.ch
        P = lambda t, off_elems=0: ctypes.c_void_p(t.data_ptr() + off_elems * t.element_size())
        for l in range(L):
            v = _vox(dims[l])
            for n in range(N):
                ci = self.cin if l == 0 else ch[l - 1]
                src = None if l == 0 else P(ws[f'pin{l}'], n * ci * v)
                first = (x, x_strides, n) if l == 0 else None
                if l < L - 1:
                    vo = _vox(dims[l + 1])
                    self._gn_stage(ws, f'enc{l}', src, ci, ch[l], dims[l], P(ws[f'cat{l}'], n * 2 * ch[l] * v), s, first=first,
                                   pool=(P(ws[f'pin{l + 1}'], n * ch[l] * vo), dims[l + 1]))
                else:
                    self._gn_stage(ws, f'enc{l}', src, ci, ch[l], dims[l], P(ws[f'b{l}'], n * ch[l] * v), s, first=first)
        for l in range(L - 2, -1, -1):
            v, vi = _vox(dims[l]), _vox(dims[l + 1])
            wpk, bias = self.packed[f'dec{l}.up']
            nv.call('iunet_convT_fwd', self.dt, self.dim, P(ws[f'b{l + 1}']), ch[l + 1] * vi,
                    P(ws[f'cat{l}'], ch[l] * v), 2 * ch[l] * v, nv.ptr(wpk), nv.ptr(bias),
                    N, dims[l + 1][0], dims[l + 1][1], dims[l + 1][2], ch[l + 1], ch[l], s)
            for n in range(N):
                self._gn_stage(ws, f'dec{l}', P(ws[f'cat{l}'], n * 2 * ch[l] * v), 2 * ch[l], ch[l], dims[l],
                               P(ws[f'b{l}'], n * ch[l] * v), s)

    def _conv3(self, x_ptr, x_ss, y_ptr, y_ss, name, N, dims, ci, co, s, ws=None, xf=0, yf=0):
        pk, bias = self.packed[name][0], self.packed[name][1]
        if self.norm == 'group':
            lay, wpk = pk.pick(self.dim, N, dims[0], dims[1], dims[2])
            nv.call('iunet_conv3_fwd', self.dt, self.dim, x_ptr, x_ss, nv.ptr(ws['raw']), co * _vox(dims), nv.ptr(wpk), None, None,
                    N, dims[0], dims[1], dims[2], ci, co, 0, lay, s)
            self._group_norm(name, ws, y_ptr, y_ss, N, dims, co, s)
            return
        probe = self.probe if (self.probe is not None and self.probe['name'] == name) else None
        if probe is not None:                       # bench.py: HIP events around THIS layer's launch inside the real step
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._conv3_launch(x_ptr, x_ss, y_ptr, y_ss, name, N, dims, ci, co, s, ws, xf, yf)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            probe['events'].append((e0, e1, N))
            return
        self._conv3_launch(x_ptr, x_ss, y_ptr, y_ss, name, N, dims, ci, co, s, ws, xf, yf)

    def _conv3_launch(self, x_ptr, x_ss, y_ptr, y_ss, name, N, dims, ci, co, s, ws=None, xf=0, yf=0):
        pk, bias = self.packed[name][0], self.packed[name][1]
        if isinstance(pk, F8Conv):
            need = nv.lib().iunet_conv3_f8_workspace_elems(self.dim, N, dims[0], dims[1], dims[2], ci, co)      # split-K scratch
            if need > (self._f8_ws.numel() if self._f8_ws is not None else 0):
                self._f8_ws = torch.empty(need, dtype=torch.float32, device=self.device)
            nv.call('iunet_conv3_f8_fwd_q', self.dt, self.dim, x_ptr, x_ss, xf, y_ptr, y_ss, yf, nv.ptr(pk.bytes), nv.ptr(pk.scale),
                    nv.ptr(bias), N, dims[0], dims[1], dims[2], ci, co, 2, nv.ptr(self._f8_ws) if need else None, s)
            return
        assert not (xf or yf)
        lay, wpk = pk.pick(self.dim, N, dims[0], dims[1], dims[2])
        nv.call('iunet_conv3_fwd', self.dt, self.dim, x_ptr, x_ss, y_ptr, y_ss, nv.ptr(wpk), nv.ptr(bias), None,
                N, dims[0], dims[1], dims[2], ci, co, 2, lay, s)

    def infer(self, x, x_strides, N, D, H, W, logits=None, probs=None, cls=None, out_strides=None,
              divisor=1.0, accumulate=False, features_only=False):
        """Run the folded network.  `x`: any torch tensor on the device (f32/f16/bf16/u8; u8 is
        scaled by 1/255 as predict.py:30 does); `x_strides` = element strides (n, c, d, h, w).
        Outputs (all optional): logits / probs fp32 written with `out_strides` (n, c, d, h, w),
        cls uint8 [N, D*H*W]."""
        if self.packed is None:
            raise RuntimeError('Engine.load_eval() has not been called')
        g = self._graph()
        if g is not None and not features_only and self.probe is None:
            # the whole forward as one C call (csrc/net.hip: the same launches on the same operators, sequenced in C++)
            self.check_shape(D, H, W)
            return g.infer(x, x_strides, N, D, H, W, logits, probs, cls, out_strides, divisor, accumulate)
        ws = self.workspace(N, D, H, W)
        dims = ws['dims']
        es = torch.tensor([], dtype=self.act_dtype).element_size()
        s = nv.stream()
        L, ch = self.levels, self.ch
        P = lambda t, off_elems=0: ctypes.c_void_p(t.data_ptr() + off_elems * t.element_size())
        q = 1 if self.q_planes() else 0            # a / cat / pin are e4m3 planes (one byte per element: the offsets below hold as they are)
        gn_fused = self.norm == 'group' and os.environ.get('IUNET_GN_FUSED', '1') != '0'      # (0: the two-pass form, A/B switch)
        if gn_fused:
            self._infer_gn_features(ws, x, x_strides, N, s)
        for l in range(L if not gn_fused else 0):
            v = _vox(dims[l])
            if l == 0:
                w, b = self.packed['enc0.conv1'][0], self.packed['enc0.conv1'][1]
                gn = self.norm == 'group'
                if q:
                    nv.call('iunet_first_conv_fwd_q', self.dt, self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype],
                            nv.ll_array(x_strides), P(ws['a0']), ch[0] * v, nv.ptr(w), nv.ptr(b),
                            N, dims[0][0], dims[0][1], dims[0][2], self.cin, ch[0], 1, s)
                else:
                    nv.call('iunet_first_conv_fwd', self.dt, self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype],
                            nv.ll_array(x_strides), P(ws['raw']) if gn else P(ws['a0']), ch[0] * v, nv.ptr(w), nv.ptr(b), None,
                            N, dims[0][0], dims[0][1], dims[0][2], self.cin, ch[0], 0 if gn else 1, s)
                if gn:
                    self._group_norm('enc0.conv1', ws, P(ws['a0']), ch[0] * v, N, dims[0], ch[0], s)
            else:
                self._conv3(P(ws[f'pin{l}']), ch[l - 1] * v, P(ws[f'a{l}']), ch[l] * v, f'enc{l}.conv1', N, dims[l],
                            ch[l - 1], ch[l], s, ws, q, q)
            if l < L - 1:
                self._conv3(P(ws[f'a{l}']), ch[l] * v, P(ws[f'cat{l}']), 2 * ch[l] * v, f'enc{l}.conv2', N, dims[l],
                            ch[l], ch[l], s, ws, q, q)
                vo = _vox(dims[l + 1])
                if q:
                    nv.call('iunet_maxpool_q_fwd', self.dim, P(ws[f'cat{l}']), 2 * ch[l] * v, P(ws[f'pin{l + 1}']),
                            ch[l] * vo, ch[l], N, dims[l + 1][0], dims[l + 1][1], dims[l + 1][2], s)
                else:
                    nv.call('iunet_maxpool_fwd', self.dt, self.dim, P(ws[f'cat{l}']), 2 * ch[l] * v, P(ws[f'pin{l + 1}']),
                            ch[l] * vo, ch[l], N, dims[l + 1][0], dims[l + 1][1], dims[l + 1][2], s)
            else:
                self._conv3(P(ws[f'a{l}']), ch[l] * v, P(ws[f'b{l}']), ch[l] * v, f'enc{l}.conv2', N, dims[l],
                            ch[l], ch[l], s, ws, q, 0)
        for l in range(L - 2, -1, -1) if not gn_fused else ():
            v, vi = _vox(dims[l]), _vox(dims[l + 1])
            wpk, bias = self.packed[f'dec{l}.up']
            nv.call('iunet_convT_fwd_q' if q else 'iunet_convT_fwd', self.dt, self.dim, P(ws[f'b{l + 1}']), ch[l + 1] * vi,
                    P(ws[f'cat{l}'], ch[l] * v), 2 * ch[l] * v, nv.ptr(wpk), nv.ptr(bias),
                    N, dims[l + 1][0], dims[l + 1][1], dims[l + 1][2], ch[l + 1], ch[l], s)
            self._conv3(P(ws[f'cat{l}']), 2 * ch[l] * v, P(ws[f'a{l}']), ch[l] * v, f'dec{l}.conv1', N, dims[l],
                        2 * ch[l], ch[l], s, ws, q, q)
            self._conv3(P(ws[f'a{l}']), ch[l] * v, P(ws[f'b{l}']), ch[l] * v, f'dec{l}.conv2', N, dims[l],
                        ch[l], ch[l], s, ws, q, 0)
        if features_only:
            return ws['b0']                       # input of the head, NHWC8c
        hw, hb = self.packed['head']
        if out_strides is None:
            v = _vox(dims[0])
            out_strides = (self.ncls * v, v, H * W, W, 1)          # contiguous NC(D)HW
        nv.call('iunet_head_fwd', self.dt, P(ws['b0']), ch[0] * _vox(dims[0]), ch[0], nv.ptr(hw), nv.ptr(hb),
                self.ncls, nv.ptr(logits), nv.ptr(probs), nv.ptr(cls), nv.ll_array(out_strides),
                float(divisor), int(bool(accumulate)), N, D, H, W, s)

    # ------------------------------------------------------------------ layout helpers (tests / debugging)
    def to_blocked(self, t):
        """[N, C, *spatial] -> flat NHWC8c tensor in the activation dtype."""
        N, C = t.shape[:2]
        sp = t.shape[2:]
        t = t.reshape(N, C // 8, 8, *sp)
        perm = [0, 1] + list(range(3, 3 + len(sp))) + [2]
        return t.permute(*perm).contiguous().to(self.act_dtype).reshape(-1)

    def from_blocked(self, flat, N, C, sp):
        t = flat.reshape(N, C // 8, *sp, 8)
        perm = [0, 1, 2 + len(sp)] + list(range(2, 2 + len(sp)))
        return t.permute(*perm).reshape(N, C, *sp)
