"""The prediction forward as ONE C call: the handle level of the C ABI (csrc/net.hip, include/iunet.h "handle level") behind the
engines' `infer`.  The launch graph of a forward -- ~30 dependent launches -- is sequenced in C++ instead of ~30 ctypes calls from
Python: the same kernels on the same operators (bit-identical, tests/test_net_handle.py) and no Python between the launches (the
Python sequence costs ~0.15 ms of host time per 2-D forward; the 2.5-D block prediction of the reference, predict.py:79-112, is three
forwards = ~90 launches per block).

`IUNET_PY_GRAPH=1` keeps every forward on the Python-sequenced engines (A/B switch).  Engines fall back to their own sequence for
what the handle does not cover: GroupNorm, fp8 operators, `features_only`, and launches with a timing probe attached."""
import ctypes
import os

import torch

from . import _native as nv

ENABLED = not os.environ.get('IUNET_PY_GRAPH')


class NetGraph:
    """iunet_net_* handle + its device buffers (flat fp32 parameters in state_dict order, packed operators, workspaces by shape)."""

    def __init__(self, dim, levels, base, cin, ncls, mode, device, act_scale=0.0, norm='batch', groups=8):
        self.lib = nv.lib()
        self.h = ctypes.c_void_p()
        nv.call('iunet_net_create_ex', dim, levels, base, cin, ncls, mode, float(act_scale), 1 if norm == 'group' else 0, int(groups), ctypes.byref(self.h))
        self.device, self.ncls, self.cin, self.mode = device, ncls, cin, mode
        self.layout = []
        for i in range(self.lib.iunet_net_num_tensors(self.h)):
            name = ctypes.create_string_buffer(96)
            off, n = ctypes.c_longlong(), ctypes.c_longlong()
            nv.call('iunet_net_param', self.h, i, name, 96, ctypes.byref(off), ctypes.byref(n))
            self.layout.append((name.value.decode(), off.value, n.value))
        self.flat = torch.empty(self.lib.iunet_net_num_params(self.h), dtype=torch.float32, device=device)
        self.packed = torch.empty(self.lib.iunet_net_packed_bytes(self.h), dtype=torch.uint8, device=device)
        self._ws = {}
        self.loaded = False          # operators packed from the current `flat`
        self.filled = False

    def __del__(self):
        try:
            if self.h:
                self.lib.iunet_net_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_params(self, params):
        """Snapshot the parameters (one copy per tensor into the flat vector); the fold + pack launches wait for the first forward."""
        with torch.no_grad():
            ts = [params[name].detach() for name, _, _ in self.layout]
            if all(t.device == self.flat.device and t.dtype == torch.float32 for t in ts):
                torch.cat([t.reshape(-1) for t in ts], out=self.flat)            # one launch
            else:
                for (name, off, n), t in zip(self.layout, ts):
                    self.flat[off:off + n].copy_(t.reshape(-1))
        self.filled, self.loaded = True, False

    def workspace(self, N, D, H, W):
        key = (N, D, H, W)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.iunet_net_workspace_bytes(self.h, N, D, H, W)
            if nbytes <= 0:
                raise ValueError(f'spatial size {(D, H, W)}: not divisible by the network\'s pooling factor (and D == 1 in 2-D)')
            if len(self._ws) > 4:
                self._ws.clear()
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            if self.mode >= 2:
                ws[:256].zero_()           # the range flag of the split-precision forward (csrc/net.hip: ws_layout)
        return ws

    def reset_saturation(self):
        for ws in self._ws.values():
            if self.mode >= 2:
                ws[:4].zero_()

    def saturated(self):
        """modes 2 / 3: did any forward on any of this handle's workspaces store a saturated (|act_scale x activation| >= 65504) hi word?
        One small device-to-host read per workspace; the flags are cumulative since the workspace was made / reset_saturation()."""
        return self.mode >= 2 and any(int(ws[:4].view(torch.int32).item()) >= 0x7bff for ws in self._ws.values())

    def eval_step(self, x, x_strides, N, D, H, W, y, w, tdt, kind):
        """validation_step (unet.py:104-116) as one C call: -> device tensor [Loss, Dice, IoU, MCC] (overwritten by the next call)."""
        if not self.filled:
            raise RuntimeError('NetGraph.set_params() has not been called')
        s = nv.stream()
        if not self.loaded:
            nv.call('iunet_net_load', self.h, nv.ptr(self.flat), nv.ptr(self.packed), s)
            self.loaded = True
        ws = self.workspace(N, D, H, W)
        key = (N, D, H, W)
        if not hasattr(self, '_scratch'):
            self._scratch = {}
        sc = self._scratch.get(key)
        if sc is None:
            if len(self._scratch) > 4:
                self._scratch.clear()
            sc = self._scratch[key] = torch.empty(self.lib.iunet_net_eval_scratch_bytes(self.h, N, D, H, W) + 16, dtype=torch.uint8, device=self.device)
        out4 = sc[-16:].view(torch.float32)
        nv.call('iunet_net_eval_step', self.h, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(x_strides), nv.ptr(y), nv.ptr(w), tdt, kind,
                N, D, H, W, nv.ptr(ws), nv.ptr(sc), nv.ptr(out4), s)
        return out4

    def infer(self, x, x_strides, N, D, H, W, logits=None, probs=None, cls=None, out_strides=None, divisor=1.0, accumulate=False):
        if not self.filled:
            raise RuntimeError('NetGraph.set_params() has not been called')
        s = nv.stream()
        if not self.loaded:
            nv.call('iunet_net_load', self.h, nv.ptr(self.flat), nv.ptr(self.packed), s)
            self.loaded = True
        ws = self.workspace(N, D, H, W)
        if out_strides is None:
            v = D * H * W
            out_strides = (self.ncls * v, v, H * W, W, 1)
        nv.call('iunet_net_forward', self.h, nv.ptr(x), nv.IN_DTYPE_CODE[x.dtype], nv.ll_array(x_strides), N, D, H, W, nv.ptr(ws),
                nv.ptr(logits), nv.ptr(probs), nv.ptr(cls), nv.ll_array(out_strides), float(divisor), int(bool(accumulate)), s)
