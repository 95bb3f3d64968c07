"""Split-precision ("fp16x2") mode of the native forward: the tolerance-meeting path on the 16-bit matrix cores.

BASELINE.json north_star: logits within 1e-3 of the CPU fp32 path, class map integer-exact; the reference predicts in fp32
(predict.py:30-35).  Same graph and the same host interface as engine.Engine (`load_eval`, `infer`), but every activation
and every operator entry is carried as two fp16 words (hi + lo = 22 significant bits) and a product costs three
`v_mfma_f32_16x16x32_f16` instead of the sixteen-times-slower f32-input instruction of engine_f32.EngineF32
(csrc/split16.hip has the arithmetic, csrc/conv3_v4.hip the convolution).  This is what `UNet(act_dtype='fp16x2')` and the
default prediction path run.

Tensors: C channels = C/8 hi planes + C/8 lo planes of [D][H][W][8] fp16; the two halves of a skip-concat buffer stay views
([skip_hi | up_hi | skip_lo | up_lo]); activations are kept multiplied by `act_scale` (a power of two, undone exactly by the
next operator's accumulator scale).

`mixed` (the default; IUNET_X2M=0 switches it off): the stage convs evaluate the two cross terms of a split product
(x_lo w_hi, x_hi w_lo: 2^-11 of it) on the K = 128 fp8 matrix instruction -- two matrix-step units per 16 input channels instead of
three (csrc/conv3_x2m.hip).  A tensor a 3x3x3 conv reads then carries hi planes + lo8 planes (e4m3 of the residual: 1 byte per element, 3 in
all; the e4m3 image of the hi word is made in LDS by the conv's loader waves and never stored); the lo planes exist only where a
transposed conv or the head reads (the last conv of a stage).

Which of the two forms a model predicts in is decided by engine_auto.EngineAuto (a calibration of x2m against fp16x2 on one tile).
"""
import ctypes
import os

import torch

from . import _native as nv
from .engine import BN_EPS, _vox


class EngineX2:
    act_dtype = 'fp16x2'
    weight_dtype = None

    def __init__(self, dim=2, levels=4, base=32, cin=1, ncls=2, device='cuda', act_scale=64.0, mixed=None, norm='batch', groups=8):
        if norm not in ('batch', 'group'):
            raise ValueError("norm must be 'batch' or 'group'")
        if norm == 'group' and base % groups:
            raise ValueError(f'{groups} groups do not divide {base} channels')
        # norm='group': GroupNorm(groups) + ReLU after every stage conv (north_star "GroupNorm/BN").  Nothing folds: the stage convs write
        # their raw output as split words (epilogue without bias / ReLU) and csrc/gn_precise.hip normalises it with per-(sample, group)
        # statistics taken in double, writing the consumer's format (fp16x2: hi + lo planes; x2m: hi + lo8 planes).  What the BatchNorm
        # network fuses into conv epilogues -- the max-pool, the head -- runs as its own launch here (the statistics come first)
        self.norm, self.groups = norm, groups
        if dim not in (2, 3):
            raise ValueError('dim must be 2 or 3')
        if base % 32 != 0:
            raise NotImplementedError('native U-Net needs base channels to be a multiple of 32')
        if not (1 <= cin <= 4):
            raise NotImplementedError('native U-Net supports 1..4 input channels')
        if not (2 <= ncls <= 10):
            raise NotImplementedError('native U-Net supports 2..10 classes (app.py:162)')
        self.dim, self.levels, self.base, self.cin, self.ncls = dim, levels, base, cin, ncls
        self.act_scale = float(act_scale)
        # cross terms of the stage convs on the fp8 matrix cores (csrc/conv3_x2m.hip; 3-D: 4 filter columns x 32 virtual channels per
        # K = 128 instruction, 2-D: 4 taps x 32)
        self.mixed = os.environ.get('IUNET_X2M', '1') != '0' if mixed is None else bool(mixed)
        self.device = torch.device(device)
        self.ch = [base * 2 ** l for l in range(levels)]
        self.taps, self.npos = 3 ** dim, 2 ** dim
        self.packed = None
        self._ws_cache = {}
        self.probe = None          # {'name': layer, 'events': []}: timing hook of one layer's launches (bench.py)
        self.use_graph = True      # False: every forward sequenced from Python (tests compare the two)
        self._g, self._gparams, self._g_dirty, self._g_last, self._g_fwd = None, None, False, None, 0      # the C++-sequenced forward (_graph)
        nv.lib()
        # range flag: every producer of the forward raises it (atomicMax, no synchronisation) to 0x7bff when a stored hi word saturated at 65504
        self._sat = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _graph(self):
        """The C++-sequenced forward (net_graph.NetGraph) on this engine's current parameters, or None where the handle level does not
        apply (GroupNorm, fp8 operators, IUNET_PY_GRAPH=1).  The handle packs its own copy of the operators (one copy of the parameters
        + ~40 launches), so it is loaded at the SECOND forward on the same parameters: a training loop that predicts once per optimiser
        step, or a validation pass that only asks for features, never pays for it; a slice / block / volume prediction does once."""
        from . import net_graph
        self._g_fwd += 1
        if self._g_fwd < 2:
            return None
        if not net_graph.ENABLED or not self.use_graph or self.weight_dtype or not (2 <= self.levels <= 6) or self._gparams is None:
            return None
        if self.mixed and self.norm == 'group':
            return None                    # (the handle sequences GroupNorm in the fp16x2 form only: iunet_net_create_ex, mode 2)
        if self._g is None:
            self._g = net_graph.NetGraph(self.dim, self.levels, self.base, self.cin, self.ncls, 3 if self.mixed else 2, self.device, act_scale=self.act_scale,
                                         norm=self.norm, groups=self.groups)
        if self._g_dirty:
            self._g.set_params(self._gparams)
            self._g_dirty = False
            self._ws_cache.clear()         # the handle has its own workspace: the Python sequence's buffers go back to the allocator
        return self._g

    def stage_names(self):
        return [f'enc{l}' for l in range(self.levels)] + [f'dec{l}' for l in range(self.levels - 2, -1, -1)]

    def stage_io(self, prefix):
        l = int(prefix[3:])
        ci = (self.cin if l == 0 else self.ch[l - 1]) if prefix.startswith('enc') else 2 * self.ch[l]
        return ci, self.ch[l]

    # ------------------------------------------------------------------ weights
    def _source(self, params, name):
        """fp32 device tensor the preparation kernel reads: the parameter itself when it already lives on the device, otherwise a
        persistent staging copy (engine.Engine._source)."""
        t = params[name].detach()
        if t.device == self.device and t.dtype == torch.float32 and t.is_contiguous():
            return t
        st = self._stage.get(name)
        if st is None or st.shape != t.shape:
            st = self._stage[name] = torch.empty(t.shape, dtype=torch.float32, device=self.device)
        st.copy_(t)
        return st

    def load_eval(self, params):
        """Fold eval-mode BatchNorm (fp32, the oracle's operation order), scale, split and pack every operator: THREE launches over
        device-resident descriptor tables (iunet_x2_prep_batch: first conv, transposed convs and -- fp16x2 -- the stage convs;
        iunet_x2m_prep_batch: the x2m stage convs; iunet_pack_batch: the fragment orders), rebuilt only when a source tensor moves.  Every
        buffer is allocated once: a re-pack after an optimiser step is launches only (no allocation, no synchronisation).
        IUNET_X2_PREP_PER_LAYER=1: the per-layer calls the tables replace (two launches per operator; same bits: tests/test_gpu_x2m.py)."""
        if not hasattr(self, '_stage'):
            self._stage, self._bufs, self._eval_sig, self._eval_tables = {}, {}, None, None
        self._gparams, self._g_dirty, self._g_fwd = params, True, 0
        self.reset_saturation()
        dev = self.device
        lib, A = nv.lib(), self.act_scale
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
            P, d_x2, d_x2m, d_pack, per_layer = {}, [], [], [], []

            def bufs(name, n_virtual, n_packed, co):
                b = self._bufs.get(name)
                if b is None:
                    b = self._bufs[name] = (torch.empty(n_virtual, dtype=torch.float32, device=dev),
                                            torch.empty(n_packed, dtype=torch.float16, device=dev),
                                            torch.empty(co, dtype=torch.float32, device=dev), torch.empty(co, dtype=torch.float32, device=dev))
                return b
            for prefix in self.stage_names():
                ci, co = self.stage_io(prefix)
                for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                    name = f'{prefix}.conv{j}'
                    first = prefix == 'enc0' and j == 1
                    w = src[f'{name}.weight']
                    bn = [src[f'{prefix}.bn{j}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')]
                    gn = self.norm == 'group'
                    if self.mixed and not first:
                        # x2m: w_hi in the padded K16 order (3-D) / the cross-pair order (2-D: three k-groups per 32-channel step)
                        # + [w_hi8 | w_lo8] in the K128 order of the fp8 step
                        key = name + '#m'
                        bm = self._bufs.get(key)
                        pmode = 2 if self.dim == 3 else 6
                        if bm is None:
                            bm = self._bufs[key] = (torch.empty(b * a * self.taps, dtype=torch.float32, device=dev),
                                                    torch.empty(nv.pack_conv3_elems(b, a, self.taps, pmode), dtype=torch.float16, device=dev),
                                                    torch.zeros(lib.iunet_x2m_w8_bytes_nd(self.dim, b, a), dtype=torch.uint8, device=dev),
                                                    torch.empty(b, dtype=torch.float32, device=dev), torch.empty(b, dtype=torch.float32, device=dev))
                        whi, w16, w8, osc, bias = bm
                        d_x2m.append(nv.make_x2_prep_desc(w, whi, osc, bias, b, a, self.taps, 3, 0, A, A, bn=None if gn else bn, w8=w8, eps=BN_EPS))
                        d_pack.append(nv.make_desc(whi, w16, b, a, self.taps, 1 if pmode == 2 else 6, torch.float16))
                        bnq = [None] * 4 if gn else [nv.ptr(t) for t in bn]
                        per_layer.append(('iunet_x2m_prep_nd', (self.dim, nv.ptr(w), nv.ptr(whi), nv.ptr(w8), nv.ptr(osc), nv.ptr(bias), bnq[0],
                                                                bnq[1], bnq[2], bnq[3], BN_EPS, A, A, b, a)))
                        per_layer.append(('iunet_pack_conv3', (0, nv.ptr(whi), None, nv.ptr(w16), b, a, self.taps, pmode)))
                        P[name] = (w16, osc, bias, w8, bn[0], bn[1]) if gn else (w16, osc, bias, w8)
                        continue
                    pmode = lib.iunet_x2_pack_mode(self.dim)      # 2: padded K16 order (3-D); 6: compact order (2-D: the cross-pair step)
                    npk = lib.iunet_pack_first_conv_elems(b, 3 * a, self.taps) if first else nv.pack_conv3_elems(b, 3 * a, self.taps, pmode)
                    wv, dst, osc, bias = bufs(name, b * 3 * a * self.taps, npk, b)
                    kc = a if first else (16 if self.dim == 3 else 32)
                    d_x2.append(nv.make_x2_prep_desc(w, wv, osc, bias, b, a, self.taps, 0, kc, A, A, bn=None if gn else bn, eps=BN_EPS))
                    d_pack.append(nv.make_desc(wv, dst, b, 3 * a, self.taps, 2 if first else (1 if pmode == 2 else 6), torch.float16))
                    bnp = [None] * 4 if gn else [nv.ptr(t) for t in bn]
                    per_layer.append(('iunet_x2_prep', (nv.ptr(w), nv.ptr(wv), nv.ptr(osc), nv.ptr(bias), bnp[0], bnp[1], bnp[2],
                                                        bnp[3], None, BN_EPS, A, A, b, a, self.taps, 0, kc)))
                    if first:
                        per_layer.append(('iunet_pack_first_conv', (0, nv.ptr(wv), None, nv.ptr(dst), b, 3 * a, self.taps)))
                    else:
                        per_layer.append(('iunet_pack_conv3', (0, nv.ptr(wv), None, nv.ptr(dst), b, 3 * a, self.taps, pmode)))
                    P[name] = (dst, osc, bias, bn[0], bn[1]) if gn else (dst, osc, bias)
            for l in range(self.levels - 2, -1, -1):
                name = f'dec{l}.up'
                w, b0 = src[f'{name}.weight'], src[f'{name}.bias']
                ci, co = self.ch[l + 1], self.ch[l]
                wv, dst, osc, bias = bufs(name, 2 * ci * co * self.npos, 2 * ci * co * self.npos, co)
                # both words once, chunked for the LDS-resident kernel
                d_x2.append(nv.make_x2_prep_desc(w, wv, osc, bias, co, ci, self.npos, 2, lib.iunet_x2_convT_kc(ci), A, A, bias_in=b0, eps=BN_EPS))
                d_pack.append(nv.make_desc(wv, dst, co, 2 * ci, self.npos, 3, torch.float16))
                per_layer.append(('iunet_x2_prep', (nv.ptr(w), nv.ptr(wv), nv.ptr(osc), nv.ptr(bias), None, None, None, None, nv.ptr(b0), BN_EPS, A, A,
                                                    co, ci, self.npos, 2, 0)))
                per_layer.append(('iunet_pack_convT', (0, nv.ptr(wv), nv.ptr(dst), 2 * ci, co, self.npos)))
                P[name] = (dst, osc, bias)
            P['head'] = (src['head.weight'].reshape(self.ncls, self.ch[0]), src['head.bias'])
            tabs = [nv.X2PrepTable(d_x2, dev, False)] + ([nv.X2PrepTable(d_x2m, dev, True)] if d_x2m else [])
            tabs.append(nv.PackTable(d_pack, dev, sources=list(src.values())))
            self._eval_tables, self._per_layer, self._eval_sig, self.packed = tabs, per_layer, sig, P
        if os.environ.get('IUNET_X2_PREP_PER_LAYER'):
            s = nv.stream()
            for fn, args in self._per_layer:
                nv.call(fn, *args, s)
            return
        for t in self._eval_tables:
            t.run()

    # ------------------------------------------------------------------ workspace
    def level_dims(self, D, H, W):
        return [((D >> l) if self.dim == 3 else 1, H >> l, W >> l) for l in range(self.levels)]

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
            # (zero-filled once: buffers the current launch sequence skips -- b0 behind the fused head, a0 behind the one-launch first
            #  stage -- must not show up as garbage in max_stored(); the workspaces are cached, the fill is not on the hot path)
            mk = lambda c, v: torch.zeros(N * 2 * c * v, dtype=torch.float16, device=self.device)     # hi + lo planes
            mk8 = lambda c, v: torch.zeros(N * c * v, dtype=torch.uint8, device=self.device)          # lo8 planes: 1 byte per element (hi8 is made in LDS from the hi words)
            mkh = lambda c, v: torch.zeros(N * c * v, dtype=torch.float16, device=self.device)        # hi planes only
            ws = {'dims': dims}
            for l in range(self.levels):
                v = _vox(dims[l])
                if self.mixed:       # a, cat, pin: read by 3x3x3 convs only (hi + lo8); b: by a transposed conv or the head (hi + lo)
                    ws[f'a{l}'], ws[f'a{l}m'] = mkh(self.ch[l], v), mk8(self.ch[l], v)
                    ws[f'b{l}'] = mk(self.ch[l], v)
                    if l < self.levels - 1:
                        ws[f'cat{l}'], ws[f'cat{l}m'] = mkh(2 * self.ch[l], v), mk8(2 * self.ch[l], v)
                    if l > 0:
                        ws[f'pin{l}'], ws[f'pin{l}m'] = mkh(self.ch[l - 1], v), mk8(self.ch[l - 1], v)
                    continue
                ws[f'a{l}'] = mk(self.ch[l], v)
                ws[f'b{l}'] = mk(self.ch[l], v)
                if l < self.levels - 1:
                    ws[f'cat{l}'] = mk(2 * self.ch[l], v)
                if l > 0:
                    ws[f'pin{l}'] = mk(self.ch[l - 1], v)
            if self.norm == 'group':
                ws['raw'] = torch.zeros(N * 2 * max(self.ch[l] * _vox(dims[l]) for l in range(self.levels)), dtype=torch.float16, device=self.device)
                ws['gnslab'] = torch.empty(max(nv.lib().iunet_gn_precise_slab_bytes(N, self.ch[l], _vox(dims[l])) for l in range(self.levels)),
                                           dtype=torch.uint8, device=self.device)
                ws['gnsc'] = torch.empty(N * max(self.ch), dtype=torch.float32, device=self.device)
                ws['gnsh'] = torch.empty(N * max(self.ch), dtype=torch.float32, device=self.device)
            if len(self._ws_cache) > 4:
                self._ws_cache.clear()
            self._ws_cache[key] = ws
        return ws

    # ------------------------------------------------------------------ forward (inference)
    def _gn(self, name, ws, yp, y_ss, y_lo, N, d, co, s):
        """relu(group_norm(raw conv output in ws['raw'])) -> the split tensor at yp (csrc/gn_precise.hip: statistics in double)."""
        v = _vox(d)
        gamma, beta = self.packed[name][3], self.packed[name][4]
        nv.call('iunet_x2_gn_relu_fwd', nv.ptr(ws['raw']), 2 * co * v, co // 8, yp, y_ss, y_lo, nv.ptr(gamma), nv.ptr(beta), self.groups, BN_EPS,
                self.act_scale, nv.ptr(ws['gnslab']), nv.ptr(ws['gnsc']), nv.ptr(ws['gnsh']), co, N, v, nv.ptr(self._sat), s)

    def _gnm(self, name, ws, yp, y_ss, y_lo, y8p, y8_ss, N, d, co, s):
        """relu(group_norm(raw conv output in ws['raw'])) -> the x2m-format tensor (hi planes at yp, lo8 planes at y8p; y8p None: a tensor a
        transposed conv or the head reads: hi + lo planes)."""
        v = _vox(d)
        gamma, beta = self.packed[name][-2], self.packed[name][-1]
        if y8p is None:
            return nv.call('iunet_x2_gn_relu_fwd', nv.ptr(ws['raw']), 2 * co * v, co // 8, yp, y_ss, y_lo, nv.ptr(gamma), nv.ptr(beta), self.groups, BN_EPS,
                           self.act_scale, nv.ptr(ws['gnslab']), nv.ptr(ws['gnsc']), nv.ptr(ws['gnsh']), co, N, v, nv.ptr(self._sat), s)
        nv.call('iunet_x2m_gn_relu_fwd', nv.ptr(ws['raw']), 2 * co * v, co // 8, yp, y_ss, y_lo, y8p, y8_ss, nv.ptr(gamma), nv.ptr(beta), self.groups,
                BN_EPS, self.act_scale, nv.ptr(ws['gnslab']), nv.ptr(ws['gnsc']), nv.ptr(ws['gnsh']), co, N, v, nv.ptr(self._sat), s)

    def _conv3(self, name, xp, x_ss, x_lo, yp, y_ss, y_lo, N, d, ci, co, s, ws=None):
        w, osc, b = self.packed[name][:3]
        if self.norm == 'group':
            # raw output (accumulator x row scale: no bias, no ReLU) as split words, then the normalisation passes
            nv.call('iunet_x2_conv3_fwd_flag', self.dim, xp, x_ss, x_lo, nv.ptr(ws['raw']), 2 * co * _vox(d), co // 8, nv.ptr(w), nv.ptr(osc), nv.ptr(b),
                    N, d[0], d[1], d[2], ci, co, 0, nv.ptr(self._sat), s)
            return self._gn(name, ws, yp, y_ss, y_lo, N, d, co, s)
        probe = self.probe if (self.probe is not None and self.probe['name'] == name) else None
        if probe is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        nv.call('iunet_x2_conv3_fwd_flag', self.dim, xp, x_ss, x_lo, yp, y_ss, y_lo, nv.ptr(w), nv.ptr(osc), nv.ptr(b),
                N, d[0], d[1], d[2], ci, co, 2, nv.ptr(self._sat), s)
        if probe is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            probe['events'].append((e0, e1, N))

    def infer_views(self, views, D, H, W, outs):
        """ONE forward over several input views of the same spatial size: views = [(x, x_strides, n)], outs = one dict per view with
        infer's output arguments (logits / probs / cls / out_strides / divisor / accumulate).  The first conv reads each view through
        its own strides into its share of the batch, the network runs once at N = sum(n), and the last conv (with the head in its
        epilogue) writes each view's output with that view's strides, in view order -- the 2.5-D block prediction (predict.py:79-112: the
        2-D net over the slices along three axes, probabilities accumulated) as one batch of 3 S slices instead of three forwards of S:
        a third of the launches, three times the work per launch at the deep levels.  Same bits as the views run one by one (the x2m
        conv keeps one summation order per voxel whatever the batch).  x2m form; other forms: the views one by one."""
        if self.packed is None:
            raise RuntimeError('EngineX2.load_eval() has not been called')
        fuse = self.mixed and self.probe is None and self.norm == 'batch' and bool(nv.lib().iunet_x2m_head_fusable(self.ncls, self.ch[0]))
        if not fuse or len(views) == 1:
            for (x, xs, n), o in zip(views, outs):
                self.infer(x, xs, n, D, H, W, **o)
            return
        self._g_last = None
        N = sum(n for _, _, n in views)
        ws = self.workspace(N, D, H, W)
        self._infer_mixed(ws, None, None, N, D, H, W, nv.stream(), head=None, views=views, heads=outs)

    def infer(self, x, x_strides, N, D, H, W, logits=None, probs=None, cls=None, out_strides=None,
              divisor=1.0, accumulate=False, features_only=False):
        """engine.Engine.infer in split precision (same arguments and output contract)."""
        if self.packed is None:
            raise RuntimeError('EngineX2.load_eval() has not been called')
        g = self._graph()
        if g is not None and not features_only and self.probe is None:
            # the whole forward as one C call (csrc/net.hip: the same launches on the same operators, sequenced in C++)
            self.check_shape(D, H, W)
            self._g_last = (x, x_strides, N, D, H, W)
            return g.infer(x, x_strides, N, D, H, W, logits, probs, cls, out_strides, divisor, accumulate)
        self._g_last = None
        ws = self.workspace(N, D, H, W)
        dims, L, ch, s = ws['dims'], self.levels, self.ch, nv.stream()
        Pt = lambda t, planes=0, v=0: ctypes.c_void_p(t.data_ptr() + 2 * planes * v * 8)      # view starting `planes` planes in
        if self.mixed:
            # the last stage conv carries the head in its epilogue where the library has that form (2 or 3 classes on 32 channels): the
            # last activation is never written.  Same bits as conv + head (tests/test_gpu_x2m.py)
            fuse = not features_only and self.probe is None and self.norm == 'batch' and bool(nv.lib().iunet_x2m_head_fusable(self.ncls, self.ch[0]))
            head = (logits, probs, cls, out_strides, divisor, accumulate) if fuse else None
            self._infer_mixed(ws, x, x_strides, N, D, H, W, s, head=head)
            if features_only:
                return ws['b0']
            if fuse:
                return None
            return self._head(ws, N, D, H, W, logits, probs, cls, out_strides, divisor, accumulate, s)
        for l in range(L):
            d, v = dims[l], _vox(dims[l])
            c8 = ch[l] // 8
            if l == 0:
                w, osc, b = self.packed['enc0.conv1'][:3]
                gn = self.norm == 'group'
                nv.call('iunet_x2m_first_conv_fwd', self.dim, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(x_strides),
                        Pt(ws['raw'] if gn else ws['a0']), 2 * ch[0] * v, c8, None, 0, nv.ptr(w), nv.ptr(osc), nv.ptr(b), self.act_scale,
                        N, d[0], d[1], d[2], self.cin, ch[0], 0 if gn else 1, nv.ptr(self._sat), s)
                if gn:
                    self._gn('enc0.conv1', ws, Pt(ws['a0']), 2 * ch[0] * v, c8, N, d, ch[0], s)
            else:
                self._conv3(f'enc{l}.conv1', Pt(ws[f'pin{l}']), 2 * ch[l - 1] * v, ch[l - 1] // 8, Pt(ws[f'a{l}']), 2 * ch[l] * v, c8,
                            N, d, ch[l - 1], ch[l], s, ws)
            if l < L - 1:
                # skip half of the concat buffer: hi planes [0, c8), lo planes [2 c8, 3 c8)
                self._conv3(f'enc{l}.conv2', Pt(ws[f'a{l}']), 2 * ch[l] * v, c8, Pt(ws[f'cat{l}']), 4 * ch[l] * v, 2 * c8,
                            N, d, ch[l], ch[l], s, ws)
                do = dims[l + 1]
                nv.call('iunet_x2_maxpool_fwd', self.dim, Pt(ws[f'cat{l}']), 4 * ch[l] * v, 2 * c8, Pt(ws[f'pin{l + 1}']),
                        2 * ch[l] * _vox(do), c8, ch[l], N, do[0], do[1], do[2], s)
            else:
                self._conv3(f'enc{l}.conv2', Pt(ws[f'a{l}']), 2 * ch[l] * v, c8, Pt(ws[f'b{l}']), 2 * ch[l] * v, c8,
                            N, d, ch[l], ch[l], s, ws)
        for l in range(L - 2, -1, -1):
            d, v, di, vi = dims[l], _vox(dims[l]), dims[l + 1], _vox(dims[l + 1])
            c8 = ch[l] // 8
            w, osc, b = self.packed[f'dec{l}.up']
            # up half of the concat buffer: hi planes [c8, 2 c8), lo planes [3 c8, 4 c8)
            nv.call('iunet_x2m_convT_fwd', self.dim, Pt(ws[f'b{l + 1}']), 2 * ch[l + 1] * vi, ch[l + 1] // 8,
                    Pt(ws[f'cat{l}'], c8, v), 4 * ch[l] * v, 2 * c8, None, 0, nv.ptr(w), nv.ptr(osc), nv.ptr(b),
                    N, di[0], di[1], di[2], ch[l + 1], ch[l], nv.ptr(self._sat), s)
            self._conv3(f'dec{l}.conv1', Pt(ws[f'cat{l}']), 4 * ch[l] * v, 2 * c8, Pt(ws[f'a{l}']), 2 * ch[l] * v, c8,
                        N, d, 2 * ch[l], ch[l], s, ws)
            self._conv3(f'dec{l}.conv2', Pt(ws[f'a{l}']), 2 * ch[l] * v, c8, Pt(ws[f'b{l}']), 2 * ch[l] * v, c8,
                        N, d, ch[l], ch[l], s, ws)
        if features_only:
            return ws['b0']                       # input of the head: [N][hi planes | lo planes], scaled by act_scale
        self._head(ws, N, D, H, W, logits, probs, cls, out_strides, divisor, accumulate, s)

    def _head(self, ws, N, D, H, W, logits, probs, cls, out_strides, divisor, accumulate, s):
        dims, ch = ws['dims'], self.ch
        hw, hb = self.packed['head']
        if out_strides is None:
            v = _vox(dims[0])
            out_strides = (self.ncls * v, v, H * W, W, 1)
        nv.call('iunet_x2_head_fwd', nv.ptr(ws['b0']), 2 * ch[0] * _vox(dims[0]), ch[0] // 8, ch[0], nv.ptr(hw), nv.ptr(hb),
                self.act_scale, self.ncls, nv.ptr(logits), nv.ptr(probs), nv.ptr(cls), nv.ll_array(out_strides),
                float(divisor), int(bool(accumulate)), N, D, H, W, s)

    def _conv3m(self, name, xp, x_ss, x8p, x8_ss, yp, y_ss, y_lo, y8p, y8_ss, N, d, ci, co, s, pool=None):
        """3x3x3 stage conv, cross terms on the fp8 matrix cores: (hi planes, lo8 planes) -> hi planes (+ lo planes if y_lo >= 0, + m8 planes).
        pool = (hi planes, stride, lo8 planes, stride) of the half-size grid: the stage's max-pool rides in the conv's epilogue."""
        w16, osc, b, w8 = self.packed[name][:4]
        if self.norm == 'group':
            # raw output (accumulator x row scale: no bias, no ReLU) as hi + lo planes, then statistics + normalise + ReLU into the consumer's
            # format: hi + lo8 planes (y8p), lo planes where y_lo >= 0
            ws, v = self._gn_ws, _vox(d)
            nv.call('iunet_x2m_conv_fwd', self.dim, xp, x_ss, x8p, x8_ss, nv.ptr(ws['raw']), 2 * co * v, co // 8, None, 0, nv.ptr(w16), nv.ptr(w8),
                    nv.ptr(osc), nv.ptr(b), N, d[0], d[1], d[2], ci, co, 0, nv.ptr(self._sat), s)
            return self._gnm(name, ws, yp, y_ss, y_lo, y8p, y8_ss, N, d, co, s)
        probe = self.probe if (self.probe is not None and self.probe['name'] == name) else None
        if probe is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if pool is not None:
            nv.call('iunet_x2m_conv_pool_fwd', self.dim, xp, x_ss, x8p, x8_ss, yp, y_ss, y_lo, y8p, y8_ss, pool[0], pool[1], pool[2], pool[3],
                    nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), N, d[0], d[1], d[2], ci, co, 2, nv.ptr(self._sat), s)
        else:
            nv.call('iunet_x2m_conv_fwd', self.dim, xp, x_ss, x8p, x8_ss, yp, y_ss, y_lo, y8p, y8_ss, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b),
                    N, d[0], d[1], d[2], ci, co, 2, nv.ptr(self._sat), s)
        if probe is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            probe['events'].append((e0, e1, N))

    def _infer_mixed(self, ws, x, x_strides, N, D, H, W, s, head=None, views=None, heads=None):
        """The forward in the x2m form (everything but the head): a / cat / pin tensors are (hi planes, lo8 planes), b tensors (hi, lo).
        views / heads (infer_views): several (x, strides, n) inputs filling the batch, and one fused-head output per view."""
        dims, L, ch = ws['dims'], self.levels, self.ch
        gn = self.norm == 'group'
        self._gn_ws = ws
        if views is None:
            views = [(x, x_strides, N)]
        P8 = lambda t, planes16=0, v=0: ctypes.c_void_p(t.data_ptr() + planes16 * v * 16)     # m8 view starting `planes16` 16-byte planes in
        Ph = lambda t, planes=0, v=0: ctypes.c_void_p(t.data_ptr() + planes * v * 16)         # hi view starting `planes` 8-channel planes in
        # 2-D, one input channel: the first encoder stage is ONE launch (the first conv is computed by the second conv's loader waves)
        stage0 = L > 1 and bool(nv.lib().iunet_x2m_first_stage_fusable(self.dim, self.cin, ch[0], N, H, W)) and self.probe is None and len(views) == 1 and not gn
        for l in range(L):
            d, v = dims[l], _vox(dims[l])
            c = ch[l]
            if l == 0 and stage0:
                fw, fosc, fb = self.packed['enc0.conv1']
                w16, osc, b, w8 = self.packed['enc0.conv2']
                do = dims[1]
                pooled = bool(nv.lib().iunet_x2m_pool_fusable(self.dim, c))
                pool = (Ph(ws['pin1']), c * _vox(do), P8(ws['pin1m']), c * _vox(do))
                nv.call('iunet_x2m_first_stage_fwd', nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(x_strides), nv.ptr(fw), nv.ptr(fosc), nv.ptr(fb),
                        self.act_scale, Ph(ws['cat0']), 2 * c * v, -1, P8(ws['cat0m']), 2 * c * v, *(pool if pooled else (None, 0, None, 0)),
                        nv.ptr(w16), nv.ptr(w8), nv.ptr(osc), nv.ptr(b), N, d[1], d[2], nv.ptr(self._sat), s)
                if not pooled:
                    nv.call('iunet_x2m_maxpool_fwd', self.dim, Ph(ws['cat0']), 2 * c * v, P8(ws['cat0m']), 2 * c * v,
                            pool[0], pool[1], pool[2], pool[3], c, N, do[0], do[1], do[2], s)
                continue
            if l == 0:
                w, osc, b = self.packed['enc0.conv1'][:3]
                n0 = 0
                for vx, vxs, vn in views:          # each view through its own strides into its share of the batch
                    if gn:                         # raw output as hi + lo planes; normalised below
                        nv.call('iunet_x2m_first_conv_fwd', self.dim, nv.ptr(vx), nv.IN_DTYPE_CODE[vx.dtype], nv.ll_array(vxs),
                                ctypes.c_void_p(ws['raw'].data_ptr() + n0 * 2 * c * v * 2), 2 * c * v, c // 8, None, 0, nv.ptr(w), nv.ptr(osc), nv.ptr(b),
                                self.act_scale, vn, d[0], d[1], d[2], self.cin, c, 0, nv.ptr(self._sat), s)
                    else:
                        nv.call('iunet_x2m_first_conv_fwd', self.dim, nv.ptr(vx), nv.IN_DTYPE_CODE[vx.dtype], nv.ll_array(vxs),
                                ctypes.c_void_p(ws['a0'].data_ptr() + n0 * c * v * 2), c * v, -1, ctypes.c_void_p(ws['a0m'].data_ptr() + n0 * c * v),
                                c * v, nv.ptr(w), nv.ptr(osc), nv.ptr(b), self.act_scale,
                                vn, d[0], d[1], d[2], self.cin, c, 1, nv.ptr(self._sat), s)
                    n0 += vn
                if gn:
                    self._gnm('enc0.conv1', ws, Ph(ws['a0']), c * v, -1, P8(ws['a0m']), c * v, N, d, c, s)
            else:
                cp = ch[l - 1]
                self._conv3m(f'enc{l}.conv1', Ph(ws[f'pin{l}']), cp * v, P8(ws[f'pin{l}m']), cp * v, Ph(ws[f'a{l}']), c * v, -1,
                             P8(ws[f'a{l}m']), c * v, N, d, cp, c, s)
            if l < L - 1:
                # skip half of the concat buffer: hi planes [0, c / 8), lo8 planes [0, c / 16)
                do = dims[l + 1]
                fused = not gn and bool(nv.lib().iunet_x2m_pool_fusable(self.dim, c))
                pool = (Ph(ws[f'pin{l + 1}']), c * _vox(do), P8(ws[f'pin{l + 1}m']), c * _vox(do))
                self._conv3m(f'enc{l}.conv2', Ph(ws[f'a{l}']), c * v, P8(ws[f'a{l}m']), c * v, Ph(ws[f'cat{l}']), 2 * c * v, -1,
                             P8(ws[f'cat{l}m']), 2 * c * v, N, d, c, c, s, pool=pool if fused else None)
                if not fused:
                    nv.call('iunet_x2m_maxpool_fwd', self.dim, Ph(ws[f'cat{l}']), 2 * c * v, P8(ws[f'cat{l}m']), 2 * c * v,
                            pool[0], pool[1], pool[2], pool[3], c, N, do[0], do[1], do[2], s)
            else:
                self._conv3m(f'enc{l}.conv2', Ph(ws[f'a{l}']), c * v, P8(ws[f'a{l}m']), c * v, nv.ptr(ws[f'b{l}']), 2 * c * v, c // 8,
                             None, 0, N, d, c, c, s)
        for l in range(L - 2, -1, -1):
            d, v, di, vi = dims[l], _vox(dims[l]), dims[l + 1], _vox(dims[l + 1])
            c, cn = ch[l], ch[l + 1]
            w, osc, b = self.packed[f'dec{l}.up']
            # up half of the concat buffer: hi planes [c / 8, 2 c / 8), lo8 planes [c / 16, 2 c / 16)
            nv.call('iunet_x2m_convT_fwd', self.dim, nv.ptr(ws[f'b{l + 1}']), 2 * cn * vi, cn // 8, Ph(ws[f'cat{l}'], c // 8, v), 2 * c * v, -1,
                    P8(ws[f'cat{l}m'], c // 16, v), 2 * c * v, nv.ptr(w), nv.ptr(osc), nv.ptr(b), N, di[0], di[1], di[2], cn, c, nv.ptr(self._sat), s)
            self._conv3m(f'dec{l}.conv1', Ph(ws[f'cat{l}']), 2 * c * v, P8(ws[f'cat{l}m']), 2 * c * v, Ph(ws[f'a{l}']), c * v, -1,
                         P8(ws[f'a{l}m']), c * v, N, d, 2 * c, c, s)
            if l == 0 and (head is not None or heads is not None):
                w16, osc, b, w8 = self.packed['dec0.conv2']
                hw, hb = self.packed['head']
                hl = [dict(zip(('logits', 'probs', 'cls', 'out_strides', 'divisor', 'accumulate'), head))] if heads is None else heads
                n0 = 0
                for (_, _, vn), o in zip(views, hl):          # the last conv, head in its epilogue, per view: that view's output strides
                    out_strides = o.get('out_strides') or (self.ncls * v, v, H * W, W, 1)
                    nv.call('iunet_x2m_conv_head_fwd', self.dim, ctypes.c_void_p(ws['a0'].data_ptr() + n0 * c * v * 2), c * v,
                            ctypes.c_void_p(ws['a0m'].data_ptr() + n0 * c * v), c * v, nv.ptr(w16), nv.ptr(w8), nv.ptr(osc),
                            nv.ptr(b), nv.ptr(hw), nv.ptr(hb), self.act_scale, self.ncls, nv.ptr(o.get('logits')), nv.ptr(o.get('probs')),
                            nv.ptr(o.get('cls')), nv.ll_array(out_strides), float(o.get('divisor', 1.0)), int(bool(o.get('accumulate', False))),
                            vn, d[0], d[1], d[2], c, nv.ptr(self._sat), s)
                    n0 += vn
                continue
            self._conv3m(f'dec{l}.conv2', Ph(ws[f'a{l}']), c * v, P8(ws[f'a{l}m']), c * v, nv.ptr(ws[f'b{l}']), 2 * c * v, c // 8,
                         None, 0, N, d, c, c, s)

    # ------------------------------------------------------------------ range check
    def max_stored(self):
        """Largest |stored hi word| of the last forward's activations (a host-synchronising diagnostic, never on the hot path).  The
        mode keeps act_scale x activation in fp16: a value of 65504 means an activation saturated (|activation| >= 65504 / act_scale
        = 1 023 at the default 2^6) and the result is no longer within tolerance -- lower act_scale for such a model."""
        if self._g_last is not None:          # the last forward ran inside the C++ graph: replay it here so that its activations can be read
            self.infer(*self._g_last, features_only=True)
        m = 0.0
        for ws in self._ws_cache.values():
            for k, t in ws.items():
                if k != 'dims' and t.dtype == torch.float16:
                    m = max(m, float(t.abs().max()))
        return m

    def reset_saturation(self):
        """Lower the range flags (this engine's and the C++ graph's workspaces'): `saturated()` then answers for the forwards that follow.
        load_eval calls it (new weights: what an earlier network did says nothing), predict_volumes per volume."""
        self._sat.zero_()
        if self._g is not None:
            self._g.reset_saturation()

    def saturated(self):
        """Did an activation saturate (a host-synchronising diagnostic, one 4-byte read)?  EVERY forward since reset_saturation() raises
        an on-device flag in its producers' epilogues (first conv, stage convs, transposed convs; atomicMax of the saturated word's bit
        pattern, nothing on the hot path), whichever sequence -- Python or the C++ graph -- ran it: the answer covers every block of a
        volume, not the last one.  `max_stored()` still scans the last forward's tensors for the value itself."""
        return int(self._sat.item()) >= 0x7bff or (self._g is not None and self._g.saturated())

    # ------------------------------------------------------------------ layout helpers (tests)
    def to_split(self, t):
        """fp32 [N, C, *spatial] -> flat split tensor [N][C/8 hi planes | C/8 lo planes][*spatial][8] of act_scale * t."""
        N, C = t.shape[:2]
        sp = t.shape[2:]
        v = t.float() * self.act_scale
        hi = v.to(torch.float16)
        lo = (v - hi.float()).to(torch.float16)

        def blocked(u):
            u = u.reshape(N, C // 8, 8, *sp)
            return u.permute(0, 1, *range(3, 3 + len(sp)), 2).contiguous()
        return torch.cat([blocked(hi), blocked(lo)], 1).reshape(-1)

    def from_split(self, flat, N, C, sp):
        t = flat.reshape(N, 2, C // 8, *sp, 8).float()
        t = (t[:, 0] + t[:, 1]) / self.act_scale
        return t.permute(0, 1, 2 + len(sp), *range(2, 2 + len(sp))).reshape(N, C, *sp)
