"""Drop-in for interactive_unet/unet.py: `UNet` with the reference constructor signature
(unet.py:15-20), `forward(x) -> softmax probabilities NCHW` (unet.py:65-69), `.lr`,
`.loss_function`, `.device`, `load_from_checkpoint(checkpoint_path=)`; the network itself is
the native canonical U-Net (engine.py) instead of segmentation_models_pytorch.

Only architecture='U-Net' exists natively; `encoder_name` is accepted and ignored (the canonical
net has its own plain conv encoder), `pretrained` is a no-op with a warning (no imagenet
weights for a from-scratch encoder; no network access).  Extra keyword arguments (dim, levels,
base, act_dtype, infer_dtype) select the 3-D / wider variants of BASELINE.json's configs.

Numeric modes.  The reference TRAINS under `precision='16-mixed'` (trainer.py:59) and PREDICTS in fp32 (predict.py:30-35).
The default module (act_dtype=None) does the same: native training with fp16 activations, `forward()` / prediction in the
split-precision mode 'fp16x2' (logits within 1e-3 of the CPU fp32 path) -- engine_auto.EngineAuto picks, by a calibration on
the model's own weights and input, the faster x2m form (cross terms on the fp8 matrix cores) where it holds the tolerance with
margin and the full fp16x2 form (engine_x2.py: every value two fp16 words, three 16-bit MFMAs per product) where it does not,
and re-runs a prediction whose activations left the fp16 range in a wider form (`infer_policy` = 'x2m' / 'fp16x2' pins one
form).  An explicit act_dtype selects one mode for both: 'fp16' / 'bf16' = 16-bit
activations on the matrix cores (the throughput path: logits off the fp32 path by 4e-3 / 3e-2); 'fp32' = the f32-input matrix
instruction for prediction (engine_f32.py) AND training (train_engine_f32.py: the parity form of the step, gradients within 1e-4 of
CPU autograd), 1/16 of the 16-bit rate; 'fp16x2' is inference-only.  `infer_dtype` overrides the
mode of `forward()` alone.
"""
import math
import warnings

import torch
import torch.nn as nn

from . import metrics
from .engine import Engine, BN_EPS

X2 = 'fp16x2'          # split precision (engine_x2.py): not a torch dtype
_ACT = {'fp16': torch.float16, 'f16': torch.float16, 'bf16': torch.bfloat16, 'fp32': torch.float32, 'f32': torch.float32,
        torch.float16: torch.float16, torch.bfloat16: torch.bfloat16, torch.float32: torch.float32, X2: X2, 'x2': X2}
_ACT_NAME = {torch.float16: 'fp16', torch.bfloat16: 'bf16', torch.float32: 'fp32', X2: X2}


def param_shapes(dim=2, levels=4, base=32, cin=1, ncls=2):
    """Ordered {name: shape} of the canonical network (same names as the oracle's definition)."""
    ch = [base * 2 ** l for l in range(levels)]
    k3, k2, k1 = (3,) * dim, (2,) * dim, (1,) * dim
    shapes = {}

    def stage(prefix, ci, co):
        for j, (a, b) in enumerate(((ci, co), (co, co)), 1):
            shapes[f'{prefix}.conv{j}.weight'] = (b, a) + k3
            for k in ('weight', 'bias', 'running_mean', 'running_var'):
                shapes[f'{prefix}.bn{j}.{k}'] = (b,)
    for l in range(levels):
        stage(f'enc{l}', cin if l == 0 else ch[l - 1], ch[l])
    for l in range(levels - 2, -1, -1):
        shapes[f'dec{l}.up.weight'] = (ch[l + 1], ch[l]) + k2
        shapes[f'dec{l}.up.bias'] = (ch[l],)
        stage(f'dec{l}', 2 * ch[l], ch[l])
    shapes['head.weight'] = (ncls, ch[0]) + k1
    shapes['head.bias'] = (ncls,)
    return shapes


def _is_buffer(name):
    return name.endswith('running_mean') or name.endswith('running_var')


class _NativeStep(torch.autograd.Function):
    """One native training step as an autograd node: forward = TrainEngine.step_forward (forward with batch statistics + fused
    head / softmax / loss), backward = TrainEngine.step_backward; the parameters are inputs so that `loss.backward()` leaves
    their gradients in `.grad` exactly as torch autograd would."""

    @staticmethod
    def forward(ctx, te, X, y, w, *params):
        out4, state = te.step_forward(X, y, w)
        ctx.te, ctx.state = te, state
        out4 = out4.clone()
        ctx.mark_non_differentiable(out4)
        return out4[0].clone(), out4

    @staticmethod
    def backward(ctx, g_loss, _g_metrics):
        te = ctx.te
        flat, _ok = te.step_backward(ctx.state)
        flat = flat * g_loss
        grads = []
        for n in te.names:
            off, sz = te.offsets[n]
            grads.append(flat[off:off + sz].view(te.p(n).shape))
        return (None, None, None, None) + tuple(grads)


class UNet(nn.Module):
    """The UNet model (native MI355X path)."""

    def __init__(self, lr=0.0001, num_channels=1, num_classes=2, loss_function=metrics.mcc_ce_loss,
                 architecture='U-Net', encoder_name='mit_b0', pretrained=True,
                 dim=2, levels=4, base=32, act_dtype=None, weight_dtype=None, norm='batch', groups=8, infer_dtype=None,
                 act_quant=None, infer_policy=None):
        super().__init__()
        if architecture != 'U-Net':
            raise NotImplementedError(f"architecture {architecture!r}: only 'U-Net' has a native MI355X "
                                      f"implementation (the reference builds the others through smp, unet.py:33-54)")
        if pretrained:
            warnings.warn('pretrained=True ignored: the native U-Net encoder is trained from scratch')
        self.hparams = dict(lr=lr, num_channels=num_channels, num_classes=num_classes,
                            loss_function=getattr(loss_function, '__name__', str(loss_function)),
                            architecture=architecture, encoder_name=encoder_name, pretrained=pretrained,
                            dim=dim, levels=levels, base=base,
                            act_dtype=None if act_dtype is None else _ACT_NAME[_ACT[act_dtype]],
                            weight_dtype=weight_dtype, norm=norm, groups=groups,
                            infer_dtype=None if infer_dtype is None else _ACT_NAME[_ACT[infer_dtype]], act_quant=act_quant,
                            infer_policy=infer_policy)
        self.lr = lr
        self.loss_function = loss_function
        self.dim, self.levels, self.base = dim, levels, base
        self.num_channels, self.num_classes = num_channels, num_classes
        # act_dtype None = the reference's pair: 16-bit training (trainer.py:59) + tolerance-meeting prediction (predict.py:30-35);
        # fp8-weight networks predict in their own mode
        self.act_dtype = torch.float16 if act_dtype is None else _ACT[act_dtype]
        if infer_dtype is not None:
            self.infer_dtype = _ACT[infer_dtype]
        elif act_dtype is None and weight_dtype is None:
            self.infer_dtype = X2              # (GroupNorm networks too: engine_auto runs them in the full fp16x2 form)
        else:
            self.infer_dtype = self.act_dtype
        # 'fp8_e4m3' (BASELINE config C5): inference runs on weights quantised to OCP e4m3 with per-output-channel
        # power-of-two scales (after the BatchNorm fold); training keeps fp32 masters and 16-bit operators.  act_quant (default
        # True = W8A8): the stage convs run on the fp8 matrix cores, which also rounds their ACTIVATIONS to e4m3; False = W8A16,
        # e4m3-valued operators on the 16-bit matrix cores with unquantised activations (engine.Engine has the numbers)
        self.weight_dtype, self.act_quant = weight_dtype, act_quant
        # form of the split-precision prediction: None / 'auto' = calibrated choice (engine_auto.py), 'x2m' / 'fp16x2' = pinned
        self.infer_policy = infer_policy
        # norm='group': GroupNorm(groups) instead of BatchNorm after every stage conv (north_star "GroupNorm/BN"); the
        # bn{j}.weight / .bias parameters are its affine pair, the running statistics are unused
        if norm not in ('batch', 'group'):
            raise ValueError("norm must be 'batch' or 'group'")
        self.norm, self.groups = norm, groups
        self._names = []
        for name, shp in param_shapes(dim, levels, base, num_channels, num_classes).items():
            t = torch.empty(shp, dtype=torch.float32)
            key = name.replace('.', '__')
            if _is_buffer(name):
                self.register_buffer(key, t)
            else:
                self.register_parameter(key, nn.Parameter(t))
            self._names.append(name)
        self.reset_parameters()
        self._engines = {}
        self._packed_sig = None
        self.logged_metrics = {}

    # ---- parameters ---------------------------------------------------------------------
    def reset_parameters(self, seed=None):
        g = None if seed is None else torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name in self._names:
                t = self.tensor(name)
                shp = t.shape
                if name.endswith('conv1.weight') or name.endswith('conv2.weight') or name == 'head.weight':
                    fan_in = shp[1] * math.prod(shp[2:])
                    t.copy_(torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_in))      # He-normal
                elif name.endswith('up.weight'):
                    t.copy_(torch.randn(shp, generator=g) * math.sqrt(1.0 / shp[0]))
                elif name.endswith('running_var') or name.endswith('bn1.weight') or name.endswith('bn2.weight'):
                    t.fill_(1.0)
                else:
                    t.zero_()

    def tensor(self, name):
        return getattr(self, name.replace('.', '__'))

    def named_tensors(self):
        return {n: self.tensor(n) for n in self._names}

    def load_named(self, tensors):
        """Copy {canonical name: tensor} (e.g. the oracle's init_params) into the module."""
        with torch.no_grad():
            for n in self._names:
                self.tensor(n).copy_(tensors[n])

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        return type(sd)((key.replace('__', '.'), v) for key, v in sd.items())

    def load_state_dict(self, sd, strict=True):
        return super().load_state_dict({k.replace('.', '__'): v for k, v in sd.items()}, strict=strict)

    @property
    def device(self):
        return self.tensor(self._names[0]).device

    # ---- engines ------------------------------------------------------------------------
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in (self.tensor(n) for n in self._names))

    def engine(self, mode='eval'):
        """The native engine with current weights packed (re-packs when a parameter changed)."""
        dev = self.device
        if dev.type != 'cuda':
            raise RuntimeError('the native U-Net runs on the GPU only: move the module with .to("cuda") '
                               '(there is no CPU fallback)')
        eng = self._engines.get(dev)
        if eng is None:
            if self.infer_dtype in (torch.float32, X2):
                if self.weight_dtype is not None:
                    raise ValueError(f"{_ACT_NAME[self.infer_dtype]!r} is a parity mode: no weight_dtype")
                if self.infer_dtype == X2:
                    from .engine_auto import EngineAuto
                    eng = EngineAuto(self.dim, self.levels, self.base, self.num_channels, self.num_classes, dev,
                                     policy=getattr(self, 'infer_policy', None), norm=self.norm, groups=self.groups)
                else:
                    from .engine_f32 import EngineF32
                    eng = EngineF32(self.dim, self.levels, self.base, self.num_channels, self.num_classes, dev, norm=self.norm,
                                    groups=self.groups)
            else:
                eng = Engine(self.dim, self.levels, self.base, self.num_channels, self.num_classes, self.infer_dtype, dev,
                             weight_dtype=self.weight_dtype, norm=self.norm, groups=self.groups, act_quant=self.act_quant)
            self._engines = {dev: eng}
            self._packed_sig = None
        sig = self._signature()
        if self._packed_sig != sig:
            eng.load_eval(self.named_tensors())
            self._packed_sig = sig
        return eng

    # ---- forward (unet.py:65-69) ----------------------------------------------------------
    def forward(self, x):
        """x [N, C, H, W] (or [N, C, D, H, W] for dim=3), float or uint8 -> softmax probabilities fp32."""
        eng = self.engine('eval')
        x = x.to(self.device)
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.uint8):
            x = x.float()
        x = x.contiguous()
        N = x.shape[0]
        sp = tuple(x.shape[2:])
        D, H, W = sp if self.dim == 3 else (1,) + sp
        vox = D * H * W
        probs = torch.empty((N, self.num_classes) + sp, dtype=torch.float32, device=self.device)
        run = lambda: eng.infer(x, (self.num_channels * vox, vox, H * W, W, 1), N, D, H, W, probs=probs)
        if hasattr(eng, 'run_checked'):
            eng.run_checked(run)          # split precision: one 4-byte read of the range flag; a saturated forward is re-run wider
        else:
            run()
        return probs

    # ---- optimiser / steps (unet.py:71-116) -------------------------------------------------
    def configure_optimizers(self):
        """unet.py:71-73: `torch.optim.AdamW(self.parameters(), lr=self.lr)` (torch defaults: betas (0.9, 0.999), eps 1e-8,
        weight_decay 1e-2).  It steps the module's parameters in place; the native engine re-packs its operators when it sees
        their version counters move.  (trainer.train_model uses the fused flat AdamW of train_engine.TrainEngine instead: the
        same update in one launch.)"""
        return torch.optim.AdamW(self.parameters(), lr=self.lr)

    def train_engine(self):
        """The native training engine bound to this module (created on first use: it re-homes the parameters in one flat tensor)."""
        te = getattr(self, '_train_engine', None)
        if te is None or te.dev != self.device:
            from .train_engine_f32 import make_train_engine
            lk = getattr(self.loss_function, 'native_kind', None)
            if lk is None:
                raise NotImplementedError(f'loss_function {self.loss_function!r} has no native kernel: use one of metrics.py\'s seven')
            te = make_train_engine(self, lr=self.lr, loss_kind=lk)       # act_dtype='fp32': the fp32 parity form of the step
            object.__setattr__(self, '_train_engine', te)
        return te

    def log(self, name, value, **kwargs):
        """Lightning's self.log (unet.py:83-86) without Lightning: the last value per name, kept as device tensors (no sync)."""
        self.logged_metrics[name] = value

    def _log_metrics(self, set_name, out4):
        # unet.py:75-86: Loss + Dice / IoU / MCC on the ROUNDED tensors -- the fused head + loss kernel computes all four
        for i, k in enumerate(('Loss', 'Dice', 'IoU', 'MCC')):
            self.log(f'{set_name}/{k}', out4[i], on_step=False, on_epoch=True)

    def training_step(self, batch, batch_idx=None):
        """unet.py:88-102: X, y, w = batch; y_hat = self(X) with BatchNorm batch statistics; loss = loss_function(y_hat, y, w,
        axes=[0, 2, 3]); metrics logged.  Returns the loss as a tensor whose `.backward()` runs the native backward pass and
        leaves the gradients in the parameters' `.grad` (so `configure_optimizers()`' AdamW, or any torch optimiser, steps them)."""
        X, y, w = batch
        te = self.train_engine()
        params = [te.p(n) for n in te.names]
        loss, out4 = _NativeStep.apply(te, X, y, w, *params)
        self._log_metrics('train', out4)
        return loss

    def validation_step(self, batch, batch_idx=None):
        """unet.py:104-116: the same with eval-mode BatchNorm (running statistics), no gradient."""
        X, y, w = batch
        te = self.train_engine()
        with torch.no_grad():
            row = te.eval_step(X, y, w)
        out4 = torch.tensor([row[k] for k in ('Loss', 'Dice', 'IoU', 'MCC')], device=self.device)
        self._log_metrics('val', out4)
        return out4[0]

    # ---- checkpoints (trainer.py:30-49, predict.py:22-24) -----------------------------------
    def save_checkpoint(self, path):
        torch.save({'state_dict': {k: v.detach().cpu() for k, v in self.state_dict().items()},
                    'hyper_parameters': dict(self.hparams)}, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **overrides):
        ck = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        hp = dict(ck['hyper_parameters'])
        hp.update(overrides)
        lf = hp.pop('loss_function', 'mcc_ce_loss')
        hp['loss_function'] = getattr(metrics, lf, metrics.mcc_ce_loss) if isinstance(lf, str) else lf
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = cls(**hp)
        model.load_state_dict(ck['state_dict'])
        if map_location is not None:
            model = model.to(map_location)
        return model
