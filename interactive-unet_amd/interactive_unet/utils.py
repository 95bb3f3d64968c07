"""Stand-alone `utils` of the native package: the names of the reference's interactive_unet/utils.py that the HOT PATH
and its direct callers dereference, so that the package imports and runs on its own (tests, bench, scripts):

* the multiscale pyramid (utils.py:18-98)            -> re-exported from `multiscale` (zoom on the device)
* `loss_name_to_function` (utils.py:458-475)          -> `metrics.loss_name_to_function` (trainer.py:28 calls it here)
* the palette and the colour <-> class conversions (utils.py:304-380) that predict.py:42, loader.py:29 / :60,
  volumedata.py:53 and suggestor.py:49 call -- plain numpy restatements (the reference's numba loop is a first-match
  palette lookup).

This file is NOT part of the drop-in: when the native modules are installed over the reference's package
(tools/install_overlay.py) the reference's own utils.py stays, because app.py:33-788 needs its
project-directory / TIFF / plotting helpers (create_directories, load_dataset, save_sample, get_training_history_figure,
...), which are GUI and file-system glue outside the hot path.  None of the native modules imports
this file.
"""
import numpy as np

from . import metrics
from .multiscale import (read_volume, resize_volume, add_multiscales, create_multiscale_zarr,     # noqa: F401
                         multiscale_levels, num_multiscale_steps)

COLORS = np.array([[0, 0, 0], [230, 25, 75], [60, 180, 75], [255, 225, 25], [0, 130, 200], [245, 130, 48],
                   [145, 30, 180], [70, 240, 240], [240, 50, 230], [210, 245, 60], [170, 255, 195]], dtype=np.uint8)   # utils.py:304-306


def loss_name_to_function(loss_function_name):
    """utils.py:458-475."""
    return metrics.loss_name_to_function(loss_function_name)


def _keys(rgb):
    rgb = np.asarray(rgb)
    return rgb[..., 0].astype(np.uint32) << 16 | rgb[..., 1].astype(np.uint32) << 8 | rgb[..., 2].astype(np.uint32)


def get_unique_colors(colored_mask):
    """utils.py:308-323: the palette colours present in the mask, in palette order."""
    return COLORS[np.isin(_keys(COLORS), _keys(colored_mask.reshape(-1, 3)))]


def colored_to_categorical(colored_mask, include_background=True):
    """utils.py:326-349: one-hot x 255 over the palette colours present (first match), then (channels 1.., weight = 255 -
    channel 0): the first present colour is the background / unlabelled channel."""
    present = get_unique_colors(colored_mask)
    keys, pk = _keys(colored_mask), _keys(present)
    mask = np.zeros(colored_mask.shape[:2] + (len(present),), dtype=np.uint8)
    for k, key in enumerate(pk):
        mask[keys == key, k] = 255
    return mask[:, :, 1:], 255 - mask[:, :, 0]


def categorical_to_colored(mask):
    """utils.py:351-357: channel i at 255 -> palette colour i + 1."""
    colored = np.zeros(mask.shape[:2] + (3,), dtype='uint8')
    for i in range(mask.shape[-1]):
        colored[mask[:, :, i] == 255, :] = COLORS[i + 1]
    return colored


def colored_to_class(colored_mask):
    """utils.py:359-368: index of the LAST non-zero categorical channel per pixel (0 where none)."""
    categorical, _ = colored_to_categorical(colored_mask)
    out = np.zeros(categorical.shape[:2], dtype='uint8')
    for i in range(categorical.shape[-1]):
        out[categorical[..., i] > 0] = i
    return out


def class_to_categorical(class_mask, num_classes, weight=None):
    """utils.py:370-380."""
    if weight is None:
        weight = np.ones(class_mask.shape)
    out = np.zeros(class_mask.shape[:2] + (num_classes,), dtype='uint8')
    for i in range(num_classes):
        out[:, :, i] = (class_mask == i) * weight
    return out
