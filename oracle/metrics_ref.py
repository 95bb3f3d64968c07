"""Oracle: restatement of the reference's soft confusion-matrix losses (numpy, float64).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows
/root/reference/interactive_unet/metrics.py line by line in meaning (not in code):

* confusion terms            metrics.py:104-166
* crossentropy_loss          metrics.py:3-21
* dice / iou / mcc           metrics.py:23-102
* X + CE combinations        metrics.py:168-187
* name -> function map       utils.py:458-475

The reference computes everything with torch ops on whatever dtype it is handed
(fp16 under autocast in training).  The oracle computes in float64 from the same
inputs so that it is the "exact" answer both the reference (goldens) and the HIP
kernel (fp32 accumulation) are compared with.

Also provides the analytic gradient d loss / d y_pred (what autograd produces in the
reference), used to pin the fused HIP loss backward.
"""
import numpy as np

EPS = 1e-12

LOSS_NAMES = {
    'Crossentropy (CE)': 'ce',
    'Dice': 'dice',
    'Intersection over Union (IoU)': 'iou',
    'Matthews correlation coefficient (MCC)': 'mcc',
    'Dice + CE': 'dice_ce',
    'IoU + CE': 'iou_ce',
    'MCC + CE': 'mcc_ce',
}
KINDS = ['ce', 'dice', 'iou', 'mcc', 'dice_ce', 'iou_ce', 'mcc_ce']


def _counts(y_true, weight, axes):
    # metrics.py:111-115: sum of weights over `axes`, or the product of those dims.
    if weight is not None:
        return np.sum(weight, axis=tuple(axes))
    return float(np.prod([y_true.shape[a] for a in axes]))


def confusion(y_pred, y_true, weight=None, axes=(2, 3)):
    """tp, tn, fp, fn fractions (metrics.py:104-166), reduced over `axes`."""
    p = np.asarray(y_pred, np.float64)
    y = np.asarray(y_true, np.float64)
    w = None if weight is None else np.asarray(weight, np.float64)
    ax = tuple(axes)
    cnt = _counts(y, w, ax)
    ww = 1.0 if w is None else w
    tp = np.sum(ww * (y * p), axis=ax) / cnt
    tn = np.sum(ww * ((1 - p) * (1 - y)), axis=ax) / cnt
    fp = np.sum(ww * ((1 - y) * p), axis=ax) / cnt
    fn = np.sum(ww * ((1 - p) * y), axis=ax) / cnt
    return tp, tn, fp, fn


def crossentropy_loss(y_pred, y_true, weight=None, axes=(2, 3)):
    p = np.asarray(y_pred, np.float64)
    y = np.asarray(y_true, np.float64)
    w = None if weight is None else np.asarray(weight, np.float64)
    ax = tuple(axes)
    ce = y * np.log(p + EPS)
    if w is not None:
        ce = w * ce
    return float(np.mean(-np.sum(ce, axis=ax) / _counts(y, w, ax)))


def dice(y_pred, y_true, weight=None, axes=(2, 3)):
    tp, tn, fp, fn = confusion(y_pred, y_true, weight, axes)
    return float(np.mean((2 * tp + EPS) / (2 * tp + fp + fn + EPS)))


def iou(y_pred, y_true, weight=None, axes=(2, 3)):
    tp, tn, fp, fn = confusion(y_pred, y_true, weight, axes)
    return float(np.mean((tp + EPS) / (tp + fp + fn + EPS)))


def mcc(y_pred, y_true, weight=None, axes=(2, 3)):
    tp, tn, fp, fn = confusion(y_pred, y_true, weight, axes)
    num = tp * tn - fp * fn
    den = ((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)) ** 0.5
    return float(np.mean((num + EPS) / (den + EPS)))


def loss(kind, y_pred, y_true, weight=None, axes=(2, 3)):
    """Any of the seven reference losses, by short kind name (see KINDS)."""
    has_ce = kind == 'ce' or kind.endswith('_ce')
    ce = crossentropy_loss(y_pred, y_true, weight, axes) if has_ce else 0.0
    base = kind[:-3] if kind.endswith('_ce') else kind
    if base == 'ce':
        return ce
    score = {'dice': dice, 'iou': iou, 'mcc': mcc}[base](y_pred, y_true, weight, axes)
    return (1.0 - score) + ce


def loss_by_name(name, *a, **k):
    return loss(LOSS_NAMES[name], *a, **k)


def loss_grad(kind, y_pred, y_true, weight=None, axes=(2, 3)):
    """d loss / d y_pred, same shape as y_pred (float64); what autograd yields in the
    reference.  Derivation: each confusion term is linear in p, so the score is a
    function of the four per-group sums; chain through them."""
    p = np.asarray(y_pred, np.float64)
    y = np.asarray(y_true, np.float64)
    w = np.ones_like(p) if weight is None else np.asarray(weight, np.float64)
    ax = tuple(axes)
    cnt = np.sum(w, axis=ax, keepdims=True) if weight is not None else _counts(y, None, ax)
    tp, tn, fp, fn = [np.expand_dims(t, ax) if np.ndim(t) else t
                      for t in confusion(p, y, weight, ax)]
    ngroups = float(np.prod([p.shape[a] for a in range(p.ndim) if a not in ax]))
    g = np.zeros_like(p)
    base = kind[:-3] if kind.endswith('_ce') else kind
    # d(term)/dp per element: tp: w*y/cnt, tn: -w*(1-y)/cnt, fp: w*(1-y)/cnt, fn: -w*y/cnt
    dtp = w * y / cnt
    dtn = -w * (1 - y) / cnt
    dfp = w * (1 - y) / cnt
    dfn = -w * y / cnt
    if base == 'dice':
        den = 2 * tp + fp + fn + EPS
        num = 2 * tp + EPS
        ds_dtp = (2 * den - 2 * num) / den ** 2
        ds_dfp = -num / den ** 2
        ds_dfn = -num / den ** 2
        g -= (ds_dtp * dtp + ds_dfp * dfp + ds_dfn * dfn) / ngroups
    elif base == 'iou':
        den = tp + fp + fn + EPS
        num = tp + EPS
        ds_dtp = (den - num) / den ** 2
        ds_dfp = -num / den ** 2
        ds_dfn = -num / den ** 2
        g -= (ds_dtp * dtp + ds_dfp * dfp + ds_dfn * dfn) / ngroups
    elif base == 'mcc':
        a, b, c, d = tp + fp, tp + fn, tn + fp, tn + fn
        prod = a * b * c * d
        root = prod ** 0.5
        num = tp * tn - fp * fn + EPS
        den = root + EPS
        # d root / d term = 0.5/root * d prod / d term   (root==0 -> inf in the reference too)
        with np.errstate(divide='ignore', invalid='ignore'):
            half = 0.5 / root
            dr_dtp = half * (b * c * d + a * c * d)
            dr_dtn = half * (a * b * d + a * b * c)
            dr_dfp = half * (b * c * d + a * b * d)
            dr_dfn = half * (a * c * d + a * b * c)
        ds_dtp = (tn * den - num * dr_dtp) / den ** 2
        ds_dtn = (tp * den - num * dr_dtn) / den ** 2
        ds_dfp = (-fn * den - num * dr_dfp) / den ** 2
        ds_dfn = (-fp * den - num * dr_dfn) / den ** 2
        g -= (ds_dtp * dtp + ds_dtn * dtn + ds_dfp * dfp + ds_dfn * dfn) / ngroups
    if kind == 'ce' or kind.endswith('_ce'):
        g += -(w * y / (p + EPS)) / cnt / ngroups
    return g


def rounded_metrics(y_pred, y_true, weight=None, axes=(0, 2, 3)):
    """Dice / IoU / MCC on round()-ed tensors, as logged by unet.py:75-86."""
    yp = np.round(np.asarray(y_pred, np.float64))
    yt = np.round(np.asarray(y_true, np.float64))
    return (dice(yp, yt, weight, axes), iou(yp, yt, weight, axes), mcc(yp, yt, weight, axes))
