"""Drop-in for interactive_unet/metrics.py: the same function names and signatures
(y_pred, y_true, weight=None, axes=[2,3]).  They serve two purposes here:

* as *loss selectors*: `UNet(loss_function=metrics.mcc_ce_loss)` / utils.loss_name_to_function
  hand one of these objects to the trainer, which maps it (attribute `native_kind`) to the fused
  HIP head+softmax+loss kernel (iunet_head_loss_fwd/bwd) -- the training path never calls them;
* as host-side utilities for logging / tests on small tensors (plain torch ops, any device).
"""
import torch

_EPS = 1e-12


def _counts(y_true, weight, axes):
    if weight is not None:
        return torch.sum(weight, dim=axes)
    n = 1
    for a in axes:
        n *= y_true.shape[a]
    return n


def _terms(y_pred, y_true, weight, axes):
    axes = list(axes)
    cnt = _counts(y_true, weight, axes)
    w = 1 if weight is None else weight
    tp = torch.sum(w * (y_true * y_pred), dim=axes) / cnt
    tn = torch.sum(w * ((1 - y_pred) * (1 - y_true)), dim=axes) / cnt
    fp = torch.sum(w * ((1 - y_true) * y_pred), dim=axes) / cnt
    fn = torch.sum(w * ((1 - y_pred) * y_true), dim=axes) / cnt
    return tp, tn, fp, fn


def true_positives(y_pred, y_true, weight=None, axes=[2, 3]):
    return _terms(y_pred, y_true, weight, axes)[0]


def true_negatives(y_pred, y_true, weight=None, axes=[2, 3]):
    return _terms(y_pred, y_true, weight, axes)[1]


def false_positives(y_pred, y_true, weight=None, axes=[2, 3]):
    return _terms(y_pred, y_true, weight, axes)[2]


def false_negatives(y_pred, y_true, weight=None, axes=[2, 3]):
    return _terms(y_pred, y_true, weight, axes)[3]


def crossentropy_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    axes = list(axes)
    ce = y_true * torch.log(y_pred + _EPS)
    if weight is not None:
        ce = weight * ce
    return torch.mean(-torch.sum(ce, dim=axes) / _counts(y_true, weight, axes))


def dice(y_pred, y_true, weight=None, axes=[2, 3]):
    tp, tn, fp, fn = _terms(y_pred, y_true, weight, axes)
    return torch.mean((2 * tp + _EPS) / (2 * tp + fp + fn + _EPS))


def iou(y_pred, y_true, weight=None, axes=[2, 3]):
    tp, tn, fp, fn = _terms(y_pred, y_true, weight, axes)
    return torch.mean((tp + _EPS) / (tp + fp + fn + _EPS))


def mcc(y_pred, y_true, weight=None, axes=[2, 3]):
    tp, tn, fp, fn = _terms(y_pred, y_true, weight, axes)
    num = tp * tn - fp * fn
    den = ((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)) ** 0.5
    return torch.mean((num + _EPS) / (den + _EPS))


def dice_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return 1 - dice(y_pred, y_true, weight, axes)


def iou_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return 1 - iou(y_pred, y_true, weight, axes)


def mcc_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return 1 - mcc(y_pred, y_true, weight, axes)


def dice_ce_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return dice_loss(y_pred, y_true, weight, axes) + crossentropy_loss(y_pred, y_true, weight, axes)


def iou_ce_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return iou_loss(y_pred, y_true, weight, axes) + crossentropy_loss(y_pred, y_true, weight, axes)


def mcc_ce_loss(y_pred, y_true, weight=None, axes=[2, 3]):
    return mcc_loss(y_pred, y_true, weight, axes) + crossentropy_loss(y_pred, y_true, weight, axes)


for _f, _k in ((crossentropy_loss, 'ce'), (dice_loss, 'dice'), (iou_loss, 'iou'), (mcc_loss, 'mcc'),
               (dice_ce_loss, 'dice_ce'), (iou_ce_loss, 'iou_ce'), (mcc_ce_loss, 'mcc_ce')):
    _f.native_kind = _k

# utils.loss_name_to_function (utils.py:458-475)
LOSS_NAMES = {
    'Crossentropy (CE)': crossentropy_loss, 'Dice': dice_loss, 'Intersection over Union (IoU)': iou_loss,
    'Matthews correlation coefficient (MCC)': mcc_loss, 'Dice + CE': dice_ce_loss, 'IoU + CE': iou_ce_loss,
    'MCC + CE': mcc_ce_loss,
}


def loss_name_to_function(loss_function_name):
    return LOSS_NAMES[loss_function_name]
