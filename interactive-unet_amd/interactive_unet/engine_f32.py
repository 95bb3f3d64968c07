"""fp32 parity mode of the native forward (BASELINE.json north_star: logits within 1e-3 of the CPU fp32 path, class
map integer-exact).  Same graph and the same host interface as engine.Engine (`load_eval`, `infer`), but fp32
activations (planar: C planes of [D][H][W]) and fp32 operators on the f32-input matrix instruction
(csrc/precise_f32.hip) -- 1/16 of the bf16 MFMA rate, so this is the checking mode, not the throughput path: it is
what `UNet(act_dtype='fp32')` runs, what the parity tests hold against oracle/unet_ref.forward_logits, and the
device-side stand-in for that oracle at sizes the CPU cannot finish (C4).

norm='group': GroupNorm(groups) + ReLU after every stage conv instead of the folded BatchNorm (north_star "GroupNorm/BN").  Nothing
folds: the conv writes its raw output, csrc/gn_precise.hip takes the per-(sample, group) statistics in double and normalises.
"""
import ctypes

import torch

from . import _native as nv
from .engine import BN_EPS, _vox


class EngineF32:
    act_dtype = torch.float32
    weight_dtype = None

    def __init__(self, dim=2, levels=4, base=32, cin=1, ncls=2, device='cuda', norm='batch', groups=8):
        if norm not in ('batch', 'group'):
            raise ValueError("norm must be 'batch' or 'group'")
        if norm == 'group' and base % groups:
            raise ValueError(f'{groups} groups do not divide {base} channels')
        self.norm, self.groups = norm, groups
        if dim not in (2, 3):
            raise ValueError('dim must be 2 or 3')
        if base % 32 != 0:
            raise NotImplementedError('native U-Net needs base channels to be a multiple of 32')
        if not (2 <= ncls <= 10):
            raise NotImplementedError('native U-Net supports 2..10 classes (app.py:162)')
        self.dim, self.levels, self.base, self.cin, self.ncls = dim, levels, base, cin, ncls
        self.device = torch.device(device)
        self.ch = [base * 2 ** l for l in range(levels)]
        self.taps, self.npos = 3 ** dim, 2 ** dim
        self.packed = None
        self._ws_cache = {}
        nv.lib()

    def stage_names(self):
        return [f'enc{l}' for l in range(self.levels)] + [f'dec{l}' for l in range(self.levels - 2, -1, -1)]

    def stage_io(self, prefix):
        l = int(prefix[3:])
        ci = (self.cin if l == 0 else self.ch[l - 1]) if prefix.startswith('enc') else 2 * self.ch[l]
        return ci, self.ch[l]

    def load_eval(self, params):
        """Fold eval-mode BatchNorm (fp32, the oracle's operation order) and pack every operator."""
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=self.device)
        src = lambda name: params[name].detach().to(self.device, torch.float32).contiguous()
        lib, s, P = nv.lib(), nv.stream(), {}
        for prefix in self.stage_names():
            ci, co = self.stage_io(prefix)
            for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                w = src(f'{prefix}.conv{j}.weight')
                bn = [src(f'{prefix}.bn{j}.{k}') for k in ('weight', 'bias', 'running_mean', 'running_var')]
                dst, bias = f32(lib.iunet_f32_pack_conv_elems(b, a, self.taps)), f32(b)
                if self.norm == 'group':          # raw operator; gamma / beta go to the normalisation pass
                    nv.call('iunet_f32_pack_conv', nv.ptr(w), nv.ptr(dst), None, None, None, None, None, BN_EPS, b, a, self.taps, 0, s)
                    P[f'{prefix}.conv{j}'] = (dst, None, bn[0], bn[1])
                    continue
                nv.call('iunet_f32_pack_conv', nv.ptr(w), nv.ptr(dst), nv.ptr(bias), nv.ptr(bn[0]), nv.ptr(bn[1]),
                        nv.ptr(bn[2]), nv.ptr(bn[3]), BN_EPS, b, a, self.taps, 0, s)
                P[f'{prefix}.conv{j}'] = (dst, bias)
        for l in range(self.levels - 2, -1, -1):
            w = src(f'dec{l}.up.weight')
            dst = f32(lib.iunet_f32_pack_conv_elems(self.ch[l], self.ch[l + 1], self.npos))
            nv.call('iunet_f32_pack_conv', nv.ptr(w), nv.ptr(dst), None, None, None, None, None, BN_EPS,
                    self.ch[l], self.ch[l + 1], self.npos, 1, s)
            P[f'dec{l}.up'] = (dst, src(f'dec{l}.up.bias'))
        P['head'] = (src('head.weight').reshape(self.ncls, self.ch[0]).contiguous(), src('head.bias'))
        torch.cuda.current_stream().synchronize()          # the staging copies above may be freed by the caller
        self.packed = P

    def level_dims(self, D, H, W):
        return [((D >> l) if self.dim == 3 else 1, H >> l, W >> l) for l in range(self.levels)]

    def workspace(self, N, D, H, W):
        key = (N, D, H, W)
        ws = self._ws_cache.get(key)
        if ws is None:
            f = 2 ** (self.levels - 1)
            if H % f or W % f or (self.dim == 3 and D % f) or (self.dim == 2 and D != 1):
                raise ValueError(f'spatial size {(D, H, W)} must be divisible by {f} (and D == 1 in 2-D)')
            dims = self.level_dims(D, H, W)
            mk = lambda c, v: torch.empty(N * c * v, dtype=torch.float32, device=self.device)
            ws = {'dims': dims}
            for l in range(self.levels):
                v = _vox(dims[l])
                ws[f'a{l}'] = mk(self.ch[l], v)
                if l < self.levels - 1:
                    ws[f'cat{l}'] = mk(2 * self.ch[l], v)
                ws[f'b{l}'] = mk(self.ch[l], v)
                if l > 0:
                    ws[f'pin{l}'] = mk(self.ch[l - 1], v)
            if self.norm == 'group':
                ws['raw'] = mk(max(self.ch[l] * _vox(dims[l]) for l in range(self.levels)), 1)
                ws['gnslab'] = torch.empty(max(nv.lib().iunet_gn_precise_slab_bytes(N, self.ch[l], _vox(dims[l])) for l in range(self.levels)),
                                           dtype=torch.uint8, device=self.device)
                ws['gnsc'], ws['gnsh'] = mk(N * max(self.ch), 1), mk(N * max(self.ch), 1)
            self._ws_cache = {key: ws}
        return ws

    def infer(self, x, x_strides, N, D, H, W, logits=None, probs=None, cls=None, out_strides=None,
              divisor=1.0, accumulate=False, features_only=False):
        """engine.Engine.infer with fp32 arithmetic end to end (same arguments and output contract)."""
        if self.packed is None:
            raise RuntimeError('EngineF32.load_eval() has not been called')
        ws = self.workspace(N, D, H, W)
        dims, L, ch, s = ws['dims'], self.levels, self.ch, nv.stream()
        Pt = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)

        def conv(name, xp, in_dt, strides, yp, y_ss, d, ci, co, transposed=0, relu=1):
            w, b = self.packed[name][0], self.packed[name][1]
            if self.norm == 'group' and not transposed:
                # raw conv output (no bias, no ReLU) -> statistics per (sample, group) in double -> relu(normalised) into the consumer's view
                gamma, beta = self.packed[name][2], self.packed[name][3]
                v = _vox(d)
                nv.call('iunet_f32_conv_fwd', self.dim, xp, in_dt, nv.ll_array(strides), nv.ptr(ws['raw']), co * v, nv.ptr(w), None,
                        N, d[0], d[1], d[2], ci, co, 0, 0, s)
                nv.call('iunet_f32_gn_relu_fwd', nv.ptr(ws['raw']), co * v, yp, y_ss, nv.ptr(gamma), nv.ptr(beta), self.groups, BN_EPS,
                        nv.ptr(ws['gnslab']), nv.ptr(ws['gnsc']), nv.ptr(ws['gnsh']), co, N, v, s)
                return
            nv.call('iunet_f32_conv_fwd', self.dim, xp, in_dt, nv.ll_array(strides), yp, y_ss, nv.ptr(w), nv.ptr(b),
                    N, d[0], d[1], d[2], ci, co, relu, transposed, s)

        planar = lambda c, d: (c * _vox(d), _vox(d), d[1] * d[2], d[2], 1)
        b_of = lambda l: ws[f'b{l}']
        for l in range(L):
            d, v = dims[l], _vox(dims[l])
            if l == 0:
                conv('enc0.conv1', nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], x_strides, Pt(ws['a0']), ch[0] * v, d, self.cin, ch[0])
            else:
                conv(f'enc{l}.conv1', Pt(ws[f'pin{l}']), 0, planar(ch[l - 1], d), Pt(ws[f'a{l}']), ch[l] * v, d, ch[l - 1], ch[l])
            if l < L - 1:
                conv(f'enc{l}.conv2', Pt(ws[f'a{l}']), 0, planar(ch[l], d), Pt(ws[f'cat{l}']), 2 * ch[l] * v, d, ch[l], ch[l])
                do = dims[l + 1]
                nv.call('iunet_f32_maxpool_fwd', self.dim, Pt(ws[f'cat{l}']), 2 * ch[l] * v, Pt(ws[f'pin{l + 1}']),
                        ch[l] * _vox(do), ch[l], N, do[0], do[1], do[2], s)
            else:
                conv(f'enc{l}.conv2', Pt(ws[f'a{l}']), 0, planar(ch[l], d), Pt(ws[f'b{l}']), ch[l] * v, d, ch[l], ch[l])
        for l in range(L - 2, -1, -1):
            d, v, di = dims[l], _vox(dims[l]), dims[l + 1]
            conv(f'dec{l}.up', Pt(b_of(l + 1)), 0, planar(ch[l + 1], di), Pt(ws[f'cat{l}'], ch[l] * v), 2 * ch[l] * v, di,
                 ch[l + 1], ch[l], transposed=1, relu=0)
            conv(f'dec{l}.conv1', Pt(ws[f'cat{l}']), 0, planar(2 * ch[l], d), Pt(ws[f'a{l}']), ch[l] * v, d, 2 * ch[l], ch[l])
            conv(f'dec{l}.conv2', Pt(ws[f'a{l}']), 0, planar(ch[l], d), Pt(b_of(l)), ch[l] * v, d, ch[l], ch[l])
        if features_only:
            return b_of(0)                          # planar fp32 [N][C0][vox]
        hw, hb = self.packed['head']
        if out_strides is None:
            v = _vox(dims[0])
            out_strides = (self.ncls * v, v, H * W, W, 1)
        nv.call('iunet_f32_head_fwd', Pt(b_of(0)), ch[0] * _vox(dims[0]), ch[0], nv.ptr(hw), nv.ptr(hb), self.ncls,
                nv.ptr(logits), nv.ptr(probs), nv.ptr(cls), nv.ll_array(out_strides), float(divisor),
                int(bool(accumulate)), N, D, H, W, s)
