"""fp32 parity form of the native training step (unet.py:88-102 + autograd + AdamW, unet.py:71-73).

`TrainEngine` (train_engine.py) trains with 16-bit activations, as the reference does under `precision='16-mixed'`
(trainer.py:59); its gradients can only be held against CPU autograd to the 16-bit storage noise.  This engine runs the SAME step
-- forward with BatchNorm batch statistics, the fused head + softmax + metrics.py loss, the full backward, the flat fused AdamW --
with planar fp32 tensors and fp32 arithmetic throughout (csrc/precise_f32.hip's f32-input MFMA convolution for every GEMM-shaped
piece: convs, their data gradients as convs with the flipped operator, transposed convs, their data gradients as 1x1 GEMMs over the
space-to-depth view; csrc/train_f32.hip for BatchNorm, pooling, the weight gradients and the head + loss), so a whole step differs
from `oracle/unet_ref.py` + torch autograd only by the order of the sums: every parameter gradient agrees to ~1e-5 relative
(tests/test_gpu_train_f32.py).  It is what `UNet(act_dtype='fp32')` trains with and the device-side checker of the 16-bit path at
sizes the CPU cannot finish; 1/16 of the 16-bit matrix rate at best -- a checking mode.

Same public surface as TrainEngine: train_step / eval_step / step_forward / step_backward / optimizer_step, `.grad`, `.flat`.
torch is used for memory, views and three layout shuffles of small tensors (operator flips, space-to-depth of one gradient).
"""
import ctypes

import torch

from . import _native as nv
from .engine import BN_EPS, _vox
from .train_engine import LOSS_KINDS

BN_MOMENTUM = 0.1


class TrainEngineF32:
    T = torch.float32
    loss_scale = 1.0

    def __init__(self, model, lr=None, loss_kind='mcc_ce', betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, process_group=None):
        self.model = model
        self.dev = model.device
        if self.dev.type != 'cuda':
            raise RuntimeError('native training runs on the GPU only (no CPU fallback)')
        if getattr(model, 'norm', 'batch') != 'batch' or getattr(model, 'weight_dtype', None):
            raise NotImplementedError('the fp32 training form covers the BatchNorm network with unquantised weights')
        nv.lib()
        self.dim, self.levels, self.ch = model.dim, model.levels, [model.base * 2 ** l for l in range(model.levels)]
        self.cin, self.ncls = model.num_channels, model.num_classes
        self.taps, self.npos = 3 ** self.dim, 2 ** self.dim
        self.lr = model.lr if lr is None else lr
        self.kind = LOSS_KINDS[loss_kind] if isinstance(loss_kind, str) else int(loss_kind)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.step_count = 0
        self.pg = process_group
        self._flatten()
        if self.pg is not None:
            from . import dp
            dp.broadcast_state(self.flat, [model.tensor(n) for n in model._names
                                           if n.endswith('running_mean') or n.endswith('running_var')], self.pg)
        self._ws, self._ops = {}, {}
        self._eval_eng = None
        self.repack()
        model._packed_sig = None

    # ------------------------------------------------------------------ parameters (train_engine.TrainEngine._flatten)
    def _flatten(self):
        m = self.model
        names = [n for n in m._names if not (n.endswith('running_mean') or n.endswith('running_var'))]
        sizes = [m.tensor(n).numel() for n in names]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.dev)
        off, self.offsets = 0, {}
        for n, s in zip(names, sizes):
            t = m.tensor(n)
            flat[off:off + s].copy_(t.detach().reshape(-1))
            t.data = flat[off:off + s].view(t.shape)
            self.offsets[n] = (off, s)
            off += s
        self.flat, self.names = flat, names
        self.grad, self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)

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

    # ------------------------------------------------------------------ operators
    def _pack(self, key, w, cout, cin, taps, transposed=0):
        """fp32 operator [cout][cin][taps] (or ConvTranspose [cin][cout][taps], transposed = 1) -> the MFMA order, cached buffer."""
        n = nv.lib().iunet_f32_pack_conv_elems(cout, cin, taps)
        buf = self._ops.get(key)
        if buf is None or buf.numel() != n:
            buf = self._ops[key] = torch.empty(n, dtype=torch.float32, device=self.dev)
        nv.call('iunet_f32_pack_conv', nv.ptr(w), nv.ptr(buf), None, None, None, None, None, BN_EPS, cout, cin, taps, transposed, nv.stream())
        return buf

    def repack(self):
        """forward and data-gradient operators of every layer from the current fp32 parameters"""
        sp = tuple(range(2, 2 + self.dim))
        keep = []
        for prefix in self.stage_names():
            ci, co, _ = self.stage_io(prefix)
            for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
                name = f'{prefix}.conv{j}'
                w = self.p(name + '.weight').detach()
                self._pack(name + '.fwd', w, b, a, self.taps)
                if a >= 32:          # data gradient: the conv with the flipped, transposed operator [cin][cout][taps] (not needed for the first conv)
                    wd = w.flip(sp).transpose(0, 1).contiguous()
                    keep.append(wd)
                    self._pack(name + '.dgrad', wd, a, b, self.taps)
        for l in range(self.levels - 2, -1, -1):
            name = f'dec{l}.up'
            w = self.p(name + '.weight').detach()                       # [cin][cout][2^d]
            ci, co = self.ch[l + 1], self.ch[l]
            self._pack(name + '.fwd', w, co, ci, self.npos, transposed=1)
            # data gradient: dx[ci][v] = sum over (pos, co) of w[ci][co][pos] dy[co][2 v + pos] -- a 1x1 GEMM over the space-to-depth view
            wd = w.reshape(ci, co, self.npos).permute(0, 2, 1).reshape(ci, self.npos * co, 1).contiguous()
            keep.append(wd)
            self._pack(name + '.dgrad', wd, ci, self.npos * co, 1)
        torch.cuda.current_stream().synchronize()      # the shuffled copies above are freed on return (a checking mode: one sync per step)

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
        f32 = lambda n: torch.empty(int(n), dtype=torch.float32, device=self.dev)
        ws = {'dims': dims}
        big = 0
        for prefix in self.stage_names():
            ci, co, l = self.stage_io(prefix)
            v = _vox(dims[l])
            for j in (1, 2):
                name = f'{prefix}.conv{j}'
                ws['y.' + name] = f32(N * co * v)
                for k in ('mean', 'std'):
                    ws[f'{k}.{name}'] = f32(co)
            ws['z1.' + prefix] = f32(N * co * v)
            ws['dz1.' + prefix] = f32(N * co * v)
            big = max(big, N * max(ci, co) * v)
        for l in range(L):
            v = _vox(dims[l])
            ws[f'b{l}'], ws[f'db{l}'] = f32(N * ch[l] * v), f32(N * ch[l] * v)
            if l < L - 1:
                ws[f'cat{l}'], ws[f'dcat{l}'] = f32(N * 2 * ch[l] * v), f32(N * 2 * ch[l] * v)
                ws[f's2d{l}'] = f32(N * ch[l] * v)                        # space-to-depth view of the up half's gradient
            if l > 0:
                ws[f'pin{l}'], ws[f'dpin{l}'] = f32(N * ch[l - 1] * v), f32(N * ch[l - 1] * v)
        ws['dy'] = f32(big)
        v0 = _vox(dims[0])
        ws['x0'] = f32(N * self.cin * v0)
        ws['dlogits'] = f32(N * self.ncls * v0)
        ws['lslab'] = f32(nv.lib().iunet_f32_head_loss_num_parts(N, v0) * self.ncls * 8)
        ws['coef'], ws['out4'] = f32(self.ncls * 3), f32(4)
        lib = nv.lib()
        need = [lib.iunet_f32_wgrad_splits(self.dim, N, *dims[0], ch[0], self.ncls) * self.ncls * ch[0]]
        for prefix in self.stage_names():
            ci, co, l = self.stage_io(prefix)
            need += [lib.iunet_f32_wgrad_splits(self.dim, N, *dims[l], a, co) * a * co * self.taps for a in (ci, co)]
        for l in range(L - 1):
            k = self.npos * ch[l]
            need.append(lib.iunet_f32_wgrad_splits(self.dim, N, *dims[l + 1], k, ch[l + 1]) * k * ch[l + 1])
        ws['wslab'] = f32(max(need))
        self._ws = {key: ws}
        return ws

    # ------------------------------------------------------------------ kernels
    @staticmethod
    def _P(t, off=0):
        return ctypes.c_void_p(t.data_ptr() + 4 * off)

    def _conv(self, op, xp, x_ss, yp, y_ss, d, ci, co, N, mode=0, bias=None):
        """planar fp32 conv (mode 0), transposed conv (1) or 1x1 conv (2) with the packed operator `op`; no ReLU"""
        v = _vox(d)
        st = nv.ll_array((x_ss, v, d[1] * d[2], d[2], 1))
        nv.call('iunet_f32_conv_fwd', self.dim, xp, 0, st, yp, y_ss, nv.ptr(self._ops[op]), nv.ptr(bias), N, d[0], d[1], d[2], ci, co,
                0, mode, nv.stream())

    def _wgrad(self, ws, xp, x_ss, dyp, dy_ss, out, d, ci, co, N, taps):
        """out (a view of the flat gradient, [co][ci][taps]) = sum over samples and voxels of dy (x) x"""
        s = nv.stream()
        splits = nv.lib().iunet_f32_wgrad_splits(self.dim, N, d[0], d[1], d[2], ci, co)
        nv.call('iunet_f32_wgrad', self.dim, xp, x_ss, dyp, dy_ss, nv.ptr(ws['wslab']), N, d[0], d[1], d[2], ci, co, taps, s)
        nv.call('iunet_reduce_slab', nv.ptr(ws['wslab']), splits, co * ci * taps, nv.ptr(out), 1.0, 0, s)

    # ------------------------------------------------------------------ forward (training mode)
    def _stage_fwd(self, ws, prefix, xp, x_ss, zp2, z2_ss, N):
        """stage = 2 x [conv -> BatchNorm (batch statistics) -> ReLU]; the second activation goes to (zp2, z2_ss)"""
        ci, co, l = self.stage_io(prefix)
        d = ws['dims'][l]
        v = _vox(d)
        s = nv.stream()
        rm = lambda n: nv.ptr(self.p(n))
        for j, (ip, i_ss, a, op_, o_ss) in enumerate(((xp, x_ss, ci, self._P(ws['z1.' + prefix]), co * v),
                                                       (self._P(ws['z1.' + prefix]), co * v, co, zp2, z2_ss)), 1):
            name = f'{prefix}.conv{j}'
            y = ws['y.' + name]
            self._conv(name + '.fwd', ip, i_ss, self._P(y), co * v, d, a, co, N)
            nv.call('iunet_f32_bn_stats', nv.ptr(y), co * v, co, N, v, BN_EPS, BN_MOMENTUM, nv.ptr(ws['mean.' + name]),
                    nv.ptr(ws['std.' + name]), rm(f'{prefix}.bn{j}.running_mean'), rm(f'{prefix}.bn{j}.running_var'), s)
            nv.call('iunet_f32_bn_relu_fwd', nv.ptr(y), co * v, op_, o_ss, nv.ptr(ws['mean.' + name]), nv.ptr(ws['std.' + name]),
                    rm(f'{prefix}.bn{j}.weight'), rm(f'{prefix}.bn{j}.bias'), co, N, v, s)

    def forward_train(self, X, N, D, H, W):
        ws = self.workspace(N, D, H, W)
        dims, L, ch, s = ws['dims'], self.levels, self.ch, nv.stream()
        v0 = _vox(dims[0])
        x0 = ws['x0'].view(N, self.cin, v0)
        x0.copy_((X.float() / 255.0 if X.dtype == torch.uint8 else X.float()).reshape(N, self.cin, v0))     # predict.py:30's scaling
        for l in range(L):
            d, v = dims[l], _vox(dims[l])
            xp, x_ss = (self._P(ws['x0']), self.cin * v) if l == 0 else (self._P(ws[f'pin{l}']), ch[l - 1] * v)
            if l < L - 1:
                self._stage_fwd(ws, f'enc{l}', xp, x_ss, self._P(ws[f'cat{l}']), 2 * ch[l] * v, N)
                do = dims[l + 1]
                nv.call('iunet_f32_maxpool_fwd', self.dim, self._P(ws[f'cat{l}']), 2 * ch[l] * v, self._P(ws[f'pin{l + 1}']),
                        ch[l] * _vox(do), ch[l], N, do[0], do[1], do[2], s)
            else:
                self._stage_fwd(ws, f'enc{l}', xp, x_ss, self._P(ws[f'b{l}']), ch[l] * v, N)
        for l in range(L - 2, -1, -1):
            d, v, di = dims[l], _vox(dims[l]), dims[l + 1]
            self._conv(f'dec{l}.up.fwd', self._P(ws[f'b{l + 1}']), ch[l + 1] * _vox(di), self._P(ws[f'cat{l}'], ch[l] * v), 2 * ch[l] * v,
                       di, ch[l + 1], ch[l], N, mode=1, bias=self.p(f'dec{l}.up.bias'))
            self._stage_fwd(ws, f'dec{l}', self._P(ws[f'cat{l}']), 2 * ch[l] * v, self._P(ws[f'b{l}']), ch[l] * v, N)
        return ws

    def loss_forward(self, ws, feat, y, w, N, vox):
        tdt = 0 if y.dtype == torch.float32 else 1
        if w is not None and w.dtype != y.dtype:
            w = w.to(y.dtype)
        nv.call('iunet_f32_head_loss_fwd', nv.ptr(feat), self.ch[0] * vox, self.ch[0], nv.ptr(self.p('head.weight')), nv.ptr(self.p('head.bias')),
                self.ncls, nv.ptr(y), nv.ptr(w), tdt, self.kind, nv.ptr(ws['lslab']), nv.ptr(ws['out4']), nv.ptr(ws['coef']), N, vox,
                nv.stream())
        return tdt, w

    # ------------------------------------------------------------------ backward
    def _stage_bwd(self, ws, prefix, dz2p, dz2_ss, xp, x_ss, dxp, dx_ss, N):
        """backward of one stage: dz2 (gradient of its output) -> parameter gradients, and the gradient of its input into (dxp, dx_ss)
        when dxp is given"""
        ci, co, l = self.stage_io(prefix)
        d = ws['dims'][l]
        v = _vox(d)
        s = nv.stream()
        z1, dz1, dy = ws['z1.' + prefix], ws['dz1.' + prefix], ws['dy']
        for j, (dzp, dz_ss, ip, i_ss, a, dip, di_ss) in ((2, (dz2p, dz2_ss, self._P(z1), co * v, co, self._P(dz1), co * v)),
                                                          (1, (self._P(dz1), co * v, xp, x_ss, ci, dxp, dx_ss))):
            name = f'{prefix}.conv{j}'
            nv.call('iunet_f32_bn_relu_bwd', dzp, dz_ss, nv.ptr(ws['y.' + name]), co * v, nv.ptr(dy), co * v, nv.ptr(ws['mean.' + name]),
                    nv.ptr(ws['std.' + name]), nv.ptr(self.p(f'{prefix}.bn{j}.weight')), nv.ptr(self.p(f'{prefix}.bn{j}.bias')),
                    nv.ptr(self.g(f'{prefix}.bn{j}.weight')), nv.ptr(self.g(f'{prefix}.bn{j}.bias')), co, N, v, s)
            self._wgrad(ws, ip, i_ss, self._P(dy), co * v, self.g(name + '.weight'), d, a, co, N, self.taps)
            if dip is not None:
                self._conv(name + '.dgrad', self._P(dy), co * v, dip, di_ss, d, co, a, N)

    def backward(self, ws, y, w, tdt, N):
        dims, L, ch, s = ws['dims'], self.levels, self.ch, nv.stream()
        v0 = _vox(dims[0])
        nv.call('iunet_f32_head_loss_bwd', self._P(ws['b0']), ch[0] * v0, ch[0], nv.ptr(self.p('head.weight')), nv.ptr(self.p('head.bias')),
                self.ncls, nv.ptr(y), nv.ptr(w), tdt, nv.ptr(ws['coef']), nv.ptr(ws['dlogits']), self.ncls * v0, self._P(ws['db0']),
                ch[0] * v0, N, v0, s)
        # head: dW[c][ch] = sum dlogits[c] x feature[ch] (a pointwise weight gradient), db[c] = sum dlogits[c]
        self._wgrad(ws, self._P(ws['b0']), ch[0] * v0, self._P(ws['dlogits']), self.ncls * v0, self.g('head.weight'), dims[0], ch[0],
                    self.ncls, N, 1)
        nv.call('iunet_f32_channel_sum', nv.ptr(ws['dlogits']), self.ncls * v0, nv.ptr(self.g('head.bias')), self.ncls, N, v0, s)
        for l in range(L - 1):                      # decoder, in reverse of the forward order
            d, v, di, vi = dims[l], _vox(dims[l]), dims[l + 1], _vox(dims[l + 1])
            self._stage_bwd(ws, f'dec{l}', self._P(ws[f'db{l}']), ch[l] * v, self._P(ws[f'cat{l}']), 2 * ch[l] * v,
                            self._P(ws[f'dcat{l}']), 2 * ch[l] * v, N)
            dup = self._P(ws[f'dcat{l}'], ch[l] * v)                     # gradient of the transposed conv's output: the up half
            nv.call('iunet_f32_channel_sum', dup, 2 * ch[l] * v, nv.ptr(self.g(f'dec{l}.up.bias')), ch[l], N, v, s)
            # space-to-depth: [N][co][2 v + pos] -> [N][pos][co][v], then the two 1x1 GEMMs over (pos, co)
            src = ws[f'dcat{l}'].view((N, 2 * ch[l]) + tuple(d))[:, ch[l]:]
            if self.dim == 3:
                s2d = src.reshape(N, ch[l], di[0], 2, di[1], 2, di[2], 2).permute(0, 3, 5, 7, 1, 2, 4, 6)
            else:
                s2d = src.reshape(N, ch[l], di[1], 2, di[2], 2).permute(0, 3, 5, 1, 2, 4)
            ws[f's2d{l}'].view(s2d.shape).copy_(s2d)
            k = self.npos * ch[l]
            self._conv(f'dec{l}.up.dgrad', self._P(ws[f's2d{l}']), k * vi, self._P(ws[f'db{l + 1}']), ch[l + 1] * vi, di, k, ch[l + 1], N, mode=2)
            # dW[ci][co][pos] = sum x[ci][v] dy'[(pos, co)][v]: the pointwise weight gradient with the roles of x and dy exchanged
            gw = self.g(f'dec{l}.up.weight')
            tmp = torch.empty(ch[l + 1] * k, dtype=torch.float32, device=self.dev)
            self._wgrad(ws, self._P(ws[f's2d{l}']), k * vi, self._P(ws[f'b{l + 1}']), ch[l + 1] * vi, tmp, di, k, ch[l + 1], N, 1)
            gw.view(ch[l + 1], ch[l], self.npos).copy_(tmp.view(ch[l + 1], self.npos, ch[l]).permute(0, 2, 1))
        for l in range(L - 1, -1, -1):              # encoder
            d, v = dims[l], _vox(dims[l])
            if l == L - 1:
                dz2p, dz2_ss = self._P(ws[f'db{l}']), ch[l] * v
            else:
                # the skip tensor's gradient: the concat's skip half + what comes back through the pool
                do = dims[l + 1]
                nv.call('iunet_f32_maxpool_bwd', self.dim, self._P(ws[f'cat{l}']), 2 * ch[l] * v, self._P(ws[f'dpin{l + 1}']), ch[l] * _vox(do),
                        self._P(ws[f'dcat{l}']), 2 * ch[l] * v, ch[l], N, do[0], do[1], do[2], 1, s)
                dz2p, dz2_ss = self._P(ws[f'dcat{l}']), 2 * ch[l] * v
            if l == 0:
                self._stage_bwd(ws, 'enc0', dz2p, dz2_ss, self._P(ws['x0']), self.cin * v, None, 0, N)
            else:
                self._stage_bwd(ws, f'enc{l}', dz2p, dz2_ss, self._P(ws[f'pin{l}']), ch[l - 1] * v, self._P(ws[f'dpin{l}']), ch[l - 1] * v, N)

    # ------------------------------------------------------------------ optimiser
    def optimizer_step(self):
        world = 1
        if self.pg is not None:
            import torch.distributed as dist
            dist.all_reduce(self.grad, group=self.pg)
            world = dist.get_world_size(self.pg)
        self.step_count += 1
        nv.call('iunet_adamw_step', nv.ptr(self.flat), nv.ptr(self.grad), nv.ptr(self.m), nv.ptr(self.v), self.flat.numel(),
                float(self.lr), self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0 / world, None, nv.stream())
        self.repack()
        self.model._packed_sig = None

    # ------------------------------------------------------------------ public steps (train_engine.TrainEngine's surface)
    def sync_weights(self):
        ver = sum(self.p(n)._version for n in self.names)
        if ver != getattr(self, '_seen_version', None):
            if getattr(self, '_seen_version', None) is not None:
                self.repack()
                self.model._packed_sig = None
            self._seen_version = ver

    def _prep(self, X, y, w):
        X = X.to(self.dev).contiguous()
        y = y.to(self.dev).contiguous()
        w = None if w is None else w.to(self.dev).contiguous()
        if y.dtype not in (torch.float16, torch.float32):
            y = y.float()
        N = X.shape[0]
        sp = tuple(X.shape[2:])
        D, H, W = sp if self.dim == 3 else (1,) + sp
        return X, y, w, N, D, H, W, D * H * W

    def step_forward(self, X, y, w=None):
        self.sync_weights()
        X, y, w, N, D, H, W, vox = self._prep(X, y, w)
        ws = self.forward_train(X, N, D, H, W)
        tdt, w = self.loss_forward(ws, ws['b0'], y, w, N, vox)
        return ws['out4'], (ws, y, w, tdt, N)

    def step_backward(self, state):
        ws, y, w, tdt, N = state
        self.backward(ws, y, w, tdt, N)
        g = self.grad.clone()
        if self.pg is not None:
            import torch.distributed as dist
            dist.all_reduce(g, group=self.pg)
            g /= dist.get_world_size(self.pg)
        return g, True

    def train_step(self, X, y, w=None, sync=True):
        out4, state = self.step_forward(X, y, w)
        ws, y, w, tdt, N = state
        self.backward(ws, y, w, tdt, N)
        self.optimizer_step()
        if sync:
            o = out4.tolist()
            return {'Loss': o[0], 'Dice': o[1], 'IoU': o[2], 'MCC': o[3]}
        return out4

    def eval_step(self, X, y, w=None, sync=True):
        """validation_step (unet.py:104-116): eval-mode BatchNorm (running statistics), fp32 forward (engine_f32.EngineF32)"""
        self.sync_weights()
        X, y, w, N, D, H, W, vox = self._prep(X, y, w)
        if X.dtype not in nv.IN_DTYPE_CODE:
            X = X.float()
        if self._eval_eng is None:
            from .engine_f32 import EngineF32
            self._eval_eng = EngineF32(self.dim, self.levels, self.model.base, self.cin, self.ncls, self.dev)
        self._eval_eng.load_eval(self.model.named_tensors())
        feat = self._eval_eng.infer(X, (self.cin * vox, vox, H * W, W, 1), N, D, H, W, features_only=True)
        ws = self.workspace(N, D, H, W)
        self.loss_forward(ws, feat, y, w, N, vox)
        if not sync:
            return ws['out4']                 # device tensor [loss, dice, iou, mcc], overwritten by the next step: clone to keep
        o = ws['out4'].tolist()
        return {'Loss': o[0], 'Dice': o[1], 'IoU': o[2], 'MCC': o[3]}


def make_train_engine(model, **kw):
    """The training engine of a module: 16-bit activations (TrainEngine) or, for act_dtype='fp32', the fp32 parity form."""
    if model.act_dtype == torch.float32:
        kw.pop('loss_scale', None)
        return TrainEngineF32(model, **kw)
    from .train_engine import TrainEngine
    return TrainEngine(model, **kw)
