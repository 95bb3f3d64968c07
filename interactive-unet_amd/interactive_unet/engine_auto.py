"""The default prediction engine: the FASTEST split-precision form that is measured to hold the north-star tolerance on THIS model.

BASELINE.json north_star: logits within 1e-3 of the CPU fp32 path; the reference predicts in fp32 (predict.py:30-35).  Two forms of the
split-precision forward exist (engine_x2.EngineX2):

* x2m    -- cross terms on the fp8 matrix cores: 0.69 of the fp16x2 time, logit error 0.5-2.5e-5 of the logit scale on trained networks
            (1-4.6e-4 absolute at logit scales 10-40, measured: tools/trained_parity.py), 3-7e-5 on random-init ones;
* fp16x2 -- every value two fp16 words: error 0.5-3e-6 of the logit scale (the fp32 modes' own sum-order noise).

x2m's margin to 1e-3 depends on the model, so nothing is assumed: EngineAuto CALIBRATES.  With new weights it runs one tile of the
caller's own input (N = 1, up to 64^3 / 256^2 voxels from the origin of the first forward's input) through BOTH forms and compares the
logits on the device (iunet_logit_diff): x2m is kept iff max |logit_x2m - logit_fp16x2| <= `threshold` (4e-4: the true x2m error is that
figure +- fp16x2's own 1e-5...6e-5, and a tile's maximum stands for a volume's with a factor 2 to spare), otherwise every forward runs in
fp16x2.  The first calibration of an engine blocks (one device-to-host read of 8 bytes); while a training loop keeps moving the weights
the calibration is repeated every `recal_every` weight loads WITHOUT a host synchronisation: the figure is copied to pinned memory
behind an event and adopted at the next weight load (the same load index on every rank of a data-parallel job: same weights, same
schedule).  `policy` / IUNET_X2M pin a form ('x2m' / '1', 'fp16x2' / '0').

Range.  Both forms keep act_scale x activation in fp16 and raise a device flag when a stored word saturates (|activation| >= 1 023 at the
default act_scale 2^6).  `run_checked(fn)` runs a prediction, reads the flag once, and on saturation RE-RUNS it one range step wider --
fp16x2 at act_scale 1 (|activation| < 65 504), then the fp32 mode (engine_f32.EngineF32) -- instead of printing a warning
(VERDICT r4 item 1c).  The wider form stays selected for the engine's lifetime.
"""
import os

import torch

from . import _native as nv
from .engine_x2 import EngineX2

THRESHOLD = 4e-4          # on max |logit_x2m - logit_fp16x2| of the calibration tile; the gate itself is 1e-3 against the CPU fp32 path
RECAL_EVERY = 16          # weight loads between two calibrations while the weights keep moving
CAL_TILE = {2: 256, 3: 64}


class EngineAuto:
    act_dtype = 'fp16x2'
    weight_dtype = None

    def __init__(self, dim=2, levels=4, base=32, cin=1, ncls=2, device='cuda', policy=None, threshold=THRESHOLD, recal_every=RECAL_EVERY,
                 norm='batch', groups=8):
        if policy is None:
            policy = {'0': 'fp16x2', '1': 'x2m'}.get(os.environ.get('IUNET_X2M', ''), 'auto')
        if policy not in ('auto', 'x2m', 'fp16x2'):
            raise ValueError("policy must be 'auto', 'x2m' or 'fp16x2'")
        # GroupNorm networks (north_star "GroupNorm/BN"): both split forms exist (csrc/gn_precise.hip writes either format), the same calibration
        self.norm, self.groups = norm, groups
        self.dim, self.levels, self.base, self.cin, self.ncls = dim, levels, base, cin, ncls
        self.device = torch.device(device)
        self.policy, self.threshold, self.recal_every = policy, float(threshold), int(recal_every)
        self.ch = [base * 2 ** l for l in range(levels)]
        self._engines = {}                 # form name -> engine
        self._loaded = {}                  # form name -> weight-load index its operators were prepared at
        self.mode = None if policy == 'auto' else policy           # form of the next forward (None: not calibrated yet -> x2m runs the calibration)
        self.wide = 0                      # range step: 0 = act_scale 2^6, 1 = fp16x2 at act_scale 1, 2 = fp32 mode
        self.calibration = None            # {'diff', 'logit_scale', 'tile', 'threshold', 'mode', 'load'} of the last adopted calibration
        self.calibrations = 0
        self._params, self._loads, self._last_cal, self._last_agree = None, 0, None, None
        self._pending = None
        self._probe = None
        nv.lib()

    # ------------------------------------------------------------------ the forms
    def _form(self, name):
        e = self._engines.get(name)
        if e is None:
            if name == 'fp32':
                from .engine_f32 import EngineF32
                e = EngineF32(self.dim, self.levels, self.base, self.cin, self.ncls, self.device, norm=self.norm, groups=self.groups)
            else:
                e = EngineX2(self.dim, self.levels, self.base, self.cin, self.ncls, self.device, mixed=(name == 'x2m'),
                             act_scale=1.0 if name == 'fp16x2_wide' else 64.0, norm=self.norm, groups=self.groups)
            self._engines[name] = e
        return e

    def _active_name(self):
        if self.wide:
            return 'fp16x2_wide' if self.wide == 1 else 'fp32'
        return self.mode or 'x2m'

    def _ready(self, name):
        """The form `name` with the current weights prepared."""
        e = self._form(name)
        if self._loaded.get(name) != self._loads:
            e.load_eval(self._params)
            self._loaded[name] = self._loads
        return e

    @property
    def active(self):
        return self._ready(self._active_name())

    @property
    def mixed(self):
        return self._active_name() == 'x2m'

    @property
    def form(self):
        """Name of the form the next forward runs in: 'x2m', 'fp16x2', 'fp16x2_wide' (act_scale 1) or 'fp32'."""
        return self._active_name()

    @property
    def packed(self):
        return self.active.packed

    @property
    def probe(self):
        return self._probe

    @probe.setter
    def probe(self, value):
        self._probe = value
        for e in self._engines.values():
            e.probe = None
        if self._params is not None:
            self.active.probe = value

    # ------------------------------------------------------------------ weights
    def load_eval(self, params):
        """New weights: adopt a calibration figure that is in flight (it was measured one load ago: the weights of a training loop move
        slowly against the rule's 2.5x margin), prepare the selected form's operators, and note whether a calibration is due."""
        self._params = params
        self._loads += 1
        if self._pending is not None:
            self._adopt()
        self._ready(self._active_name())

    def _due(self):
        if self.policy != 'auto' or self.wide:
            return False
        return self.mode is None or (self._last_cal is not None and self._loads - self._last_cal >= self.recal_every and self._pending is None)

    def collective_due(self):
        """Is a calibration AGREED OVER A PROCESS GROUP due?  A function of what every rank of a data-parallel job shares (the policy,
        the count of weight loads) and of nothing rank-local (a rank that widened its range still takes part), so that all ranks of a
        sharded prediction enter the collective together (shard.NativeOps.agree_form)."""
        return self.policy == 'auto' and (self._last_agree is None or self._loads - self._last_agree >= self.recal_every)

    # ------------------------------------------------------------------ calibration
    def _crop(self, D, H, W):
        f = 2 ** (self.levels - 1)
        t = CAL_TILE[self.dim]
        c = lambda n: max(f, min(n, t) // f * f)
        return (c(D) if self.dim == 3 else 1, c(H), c(W))

    def _measure(self, x, x_strides, D, H, W):
        """-> (device tensor [max |logit_x2m - logit_fp16x2|, max |logit_x2m|] of one tile of x -- zeros without data --, the tile's shape)."""
        out2 = torch.zeros(2, dtype=torch.float32, device=self.device)
        if x is None:
            return out2, None
        cD, cH, cW = tile = self._crop(D, H, W)
        n = self.ncls * cD * cH * cW
        lg = torch.empty(2 * n, dtype=torch.float32, device=self.device)
        for i, name in enumerate(('x2m', 'fp16x2')):
            e = self._ready(name)
            keep, e.probe, fwd = e.probe, None, e._g_fwd
            e.infer(x, x_strides, 1, cD, cH, cW, logits=lg[i * n:(i + 1) * n])
            e.probe, e._g_fwd = keep, fwd            # (a calibration forward does not count towards loading the C++ graph)
        nv.call('iunet_logit_diff', nv.ptr(lg), nv.ptr(lg[n:]), n, nv.ptr(out2), nv.stream())
        return out2, tile

    def calibrate(self, x, x_strides, D, H, W, blocking=True, group=None):
        """Run one tile (sample 0, the crop at the origin) of `x` through x2m and fp16x2 and measure max |logit difference| on the device.
        blocking: decide now (one 8-byte device-to-host read); otherwise the figure is adopted at the next load_eval.  group: a
        torch.distributed group (True = the default group) whose ranks all call this at the same point -- the figure is all-reduced (MAX)
        so that every rank of a sharded prediction takes the same decision (a rank without data passes x=None and contributes 0)."""
        if self._params is None:
            raise RuntimeError('EngineAuto.load_eval() has not been called')
        out2, tile = self._measure(x, x_strides, D, H, W)
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(out2, op=dist.ReduceOp.MAX, group=None if group is True else group)     # True: the default group
            self._last_agree = self._loads
        if out2.device.type == 'cuda':
            host = torch.empty(2, dtype=torch.float32).pin_memory()
            host.copy_(out2, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:                                   # (host tensors: the multi-process CPU tests of the agreement)
            host, ev = out2.clone(), None
        self._pending = (host, ev, tile, self._loads, out2)
        self._last_cal = self._loads
        self.calibrations += 1
        if blocking:
            self._adopt()

    def _adopt(self):
        host, ev, tile, load, _keep = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()
        diff, scale = float(host[0]), float(host[1])
        ok = diff <= self.threshold                   # (a NaN figure compares False: fp16x2)
        self.mode = 'x2m' if ok else 'fp16x2'
        self.calibration = {'diff': diff, 'logit_scale': scale, 'tile': list(tile) if tile else None, 'threshold': self.threshold,
                            'mode': self.mode, 'load': load}

    # ------------------------------------------------------------------ forward
    def infer(self, x, x_strides, N, D, H, W, logits=None, probs=None, cls=None, out_strides=None, divisor=1.0, accumulate=False,
              features_only=False):
        """engine.Engine.infer (same arguments and output contract) in the selected form; a due calibration runs first on a tile of this
        very input."""
        if self._params is None:
            raise RuntimeError('EngineAuto.load_eval() has not been called')
        if self._due():
            self.calibrate(x, x_strides, D, H, W, blocking=self.mode is None)
        e = self.active
        if getattr(e, 'probe', None) is not self._probe:
            e.probe = self._probe
        return e.infer(x, x_strides, N, D, H, W, logits=logits, probs=probs, cls=cls, out_strides=out_strides, divisor=divisor,
                       accumulate=accumulate, features_only=features_only)

    def infer_views(self, views, D, H, W, outs):
        """Several input views of one spatial size as ONE batch (engine_x2.EngineX2.infer_views: the 2.5-D block prediction's three axes in
        one forward); forms without that entry run the views one by one.  views = [(x, x_strides, n)], outs = [dict of infer's outputs]."""
        if self._params is None:
            raise RuntimeError('EngineAuto.load_eval() has not been called')
        if self._due():
            x, xs, _ = views[0]
            self.calibrate(x, xs, D, H, W, blocking=self.mode is None)
        e = self.active
        if getattr(e, 'probe', None) is not self._probe:
            e.probe = self._probe
        if hasattr(e, 'infer_views'):
            return e.infer_views(views, D, H, W, outs)
        for (x, xs, n), o in zip(views, outs):
            e.infer(x, xs, n, D, H, W, **o)

    # ------------------------------------------------------------------ range
    def saturated(self):
        e = self._engines.get(self._active_name())
        return bool(e is not None and hasattr(e, 'saturated') and e.saturated())

    def reset_saturation(self):
        for e in self._engines.values():
            if hasattr(e, 'reset_saturation'):
                e.reset_saturation()

    def max_stored(self):
        e = self._engines.get(self._active_name())
        return e.max_stored() if e is not None and hasattr(e, 'max_stored') else 0.0

    def widen(self):
        """One range step wider (fp16x2 at act_scale 1, then the fp32 mode).  -> False when there is none left."""
        if self.wide >= 2:
            return False
        self.wide += 1
        self.reset_saturation()
        return True

    def run_checked(self, fn):
        """fn() = one prediction through this engine (a slice, a block, a volume).  Reads the range flag once afterwards (a host
        synchronisation: callers that hand their result to the host pay nothing extra) and, if an activation saturated, runs fn() again
        one range step wider, until the flag stays down.  -> fn's last result."""
        self.reset_saturation()
        out = fn()
        while self.saturated():
            before = self._active_name()
            if not self.widen():
                break
            print(f'interactive_unet: an activation left the range of the {before} prediction form (|activation| x act_scale >= 65504); '
                  f'predicting again in {self._active_name()}')
            out = fn()
        return out

    def describe(self):
        """For the bench line / logs: the form in use and the calibration behind it."""
        d = {'form': self._active_name(), 'policy': self.policy, 'calibrations': self.calibrations}
        if self.calibration:
            d.update({'calibration_max_abs_logit_diff_x2m_vs_fp16x2': self.calibration['diff'], 'calibration_logit_scale': self.calibration['logit_scale'],
                      'calibration_tile': self.calibration['tile'], 'threshold': self.threshold})
        return d
