"""Drop-in for interactive_unet/suggestor.py: `Suggestor` and `make_suggestions` with the
reference's names, arguments and return values -- the 30-step on-the-fly fine-tune per brush stroke (app.py:753-778), here on
the native train step: every step is one device launch for the random flip / flip / nearest-rotation of (image, mask, weight)
(loader.UNetDataset.batch, the transforms of suggestor.py:78-81) plus TrainEngine.train_step (forward, MCC + CE loss with the
annotation mask as weight, backward, AdamW; suggestor.py:88-103), then one forward + argmax for the suggestion overlay.

Differences from the reference, by necessity: the network is the native canonical U-Net trained from scratch, not
smp.Unet('mobilenet_v2', encoder_weights='imagenet') (suggestor.py:22-26: smp and the ImageNet weights are not available, so
parity is unpinned at that boundary, as for unet.UNet).  The reference's `best_model = model.state_dict()` keeps references
to the live parameters, so its `load_state_dict(best_model)` restores nothing and the model after the last step is what
predicts; that effective behaviour is kept.
"""
import numpy as np
import torch

from . import loader, metrics, unet
from .train_engine import TrainEngine


def get_unique_colors(colored_mask):
    """utils.py:308-323: the palette colours present in the mask, in palette order."""
    flat = colored_mask.reshape(-1, 3).astype(np.uint32)
    keys = flat[:, 0] << 16 | flat[:, 1] << 8 | flat[:, 2]
    ckeys = loader.COLORS[:, 0].astype(np.uint32) << 16 | loader.COLORS[:, 1].astype(np.uint32) << 8 | loader.COLORS[:, 2]
    return loader.COLORS[np.isin(ckeys, keys)]


class Suggestor(unet.UNet):
    """suggestor.py:14-41: forward(x [B, C, H, W]) -> softmax probabilities."""

    def __init__(self, num_channels, num_classes):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            super().__init__(num_channels=num_channels, num_classes=num_classes, loss_function=metrics.mcc_ce_loss,
                             pretrained=False, dim=2)


def make_suggestions(image_features, mask, lr=0.0001, steps=30, model=None, generator=None, trace=None):
    """suggestor.py:43-116.  image_features float [1, ch, S, S] in [0, 1] (app.py:311: image / 255), mask uint8 [S, S, 3] palette
    colours (black = unlabelled) -> (suggestions uint8 [S, S, 3], model).  `trace` (a list, tests only): receives one (x, y, w, loss)
    tuple of host tensors per step -- the augmented batch the step trained on and its loss -- so that an oracle can replay the run."""
    image_size = mask.shape[0]
    unique_colors = get_unique_colors(mask)[1:]
    num_classes = len(unique_colors)
    if num_classes == 1:                              # return all same class (suggestor.py:52-54)
        return (np.ones((image_size, image_size, 3)) * unique_colors[0][None, None, :]).astype('uint8'), model
    device = torch.device('cuda', torch.cuda.current_device())
    onehot, _ = loader.colored_to_categorical(mask)
    onehot = onehot > 127                              # [S, S, C]
    x = np.asarray(image_features, dtype=np.float32)
    ch = x.shape[1]
    # the tensors of suggestor.py:60-65 as uint8 annotation planes: x = image / 255 comes back exactly, y is one-hot, w marks
    # the labelled pixels for every class
    image_u8 = np.rint(np.moveaxis(x[0], 0, -1) * 255).astype(np.uint8)
    ann = loader.annotations_from_arrays([(image_u8, onehot.astype(np.uint8) * 255, onehot.any(-1).astype(np.uint8) * 255)], device)
    ds = loader.UNetDataset(ann, None, augment=True, generator=generator, out_size=(image_size, mask.shape[1]), keep_dark=True)
    if model is None or model.num_classes != num_classes:
        model = Suggestor(ch, num_classes).to(device)
    model.train()
    engine = TrainEngine(model, lr=lr, loss_kind=metrics.mcc_ce_loss.native_kind)
    H, W = int(ann[0][0].shape[0]), int(ann[0][0].shape[1])
    for _ in range(steps):
        hflip = bool(torch.rand(1, generator=generator).item() < 0.5)              # the draws of suggestor.py:78-81, in order
        vflip = bool(torch.rand(1, generator=generator).item() < 0.5)
        angle = torch.empty(1).uniform_(-360.0, 360.0, generator=generator).item()
        xt, yt, wt = ds.batch([0], params=[(hflip, vflip, angle, (0, 0, H, W))])
        row = engine.train_step(xt, yt, wt)
        if trace is not None:
            trace.append((xt.float().cpu(), yt.float().cpu(), wt.float().cpu(), row['Loss']))
        if not np.isfinite(row['Loss']):              # suggestor.py:93-96: start over with a fresh model
            model = Suggestor(ch, num_classes).to(device)
            model.train()
            engine = TrainEngine(model, lr=lr, loss_kind=metrics.mcc_ce_loss.native_kind)
    model.eval()
    with torch.inference_mode():
        X, _, _ = loader.UNetDataset(ann, None, augment=False, keep_dark=True).batch([0])
        predictions = model(X).argmax(1)[0].cpu().numpy()
    suggestions = np.zeros((image_size, image_size, 3)).astype('uint8')
    for i in range(len(unique_colors)):
        suggestions[predictions == i, :] = unique_colors[i]
    return suggestions, model
