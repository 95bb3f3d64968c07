"""Drop-in for interactive_unet/trainer.py: `train_model(...)` with the reference's positional
signature (trainer.py:12-19; called positionally from the NiceGUI worker process,
app.py:697-719).  Lightning's Trainer.fit / ModelCheckpoint / CSVLogger are replaced by a plain
epoch loop over the native training step (train_engine.TrainEngine), keeping the files the UI
reads:

* resume from / best-val checkpoint to  model/model.ckpt            (trainer.py:30-49)
* metrics CSV at  model/history/<timestamp>/version_0/metrics.csv  (trainer.py:52; parsed by
  utils.get_training_history: columns epoch, step, train/<M>, val/<M> for Loss, Dice, IoU, MCC)
"""
import csv
import os
import time
import warnings

import torch

from . import metrics, unet
from .train_engine_f32 import make_train_engine

METRICS = ('Loss', 'Dice', 'IoU', 'MCC')


def _loaders(num_classes, batch_size, reslice, reslice_factor):
    from . import loader                           # the device batch producer (loader.py:84-101)
    tr = loader.get_data_loader(set_type='train', num_classes=num_classes, batch_size=batch_size, reslice=reslice,
                                reslice_factor=reslice_factor, augment=True, shuffle=True)
    va = loader.get_data_loader(set_type='val', num_classes=num_classes, batch_size=batch_size, reslice=False,
                                reslice_factor=reslice_factor, augment=False, shuffle=False)
    return tr, va


def _mean(rows):
    """Epoch mean of the per-step [loss, dice, iou, mcc] device tensors: ONE device-to-host transfer per epoch (a `.tolist()` per
    step kept the host from running ahead of the device: every step then paid its ~130 launches' host time in full)."""
    if not rows:
        return {k: 0.0 for k in METRICS}
    o = torch.stack(rows).double().mean(0).tolist()
    return dict(zip(METRICS, o))


def train_model(lr=0.0001, batch_size=1, epochs=10, num_channels=1, num_classes=2, loss_function_name='MCC + CE',
                architecture='U-Net', encoder_name='mit_b0', pretrained=True, reslice=False, reslice_factor=2,
                train_loader=None, val_loader=None, dim=2, act_dtype=None, process_group=None):
    if train_loader is None or val_loader is None:
        train_loader, val_loader = _loaders(num_classes, batch_size, reslice, reslice_factor)
    loss_function = metrics.loss_name_to_function(loss_function_name)
    device = torch.device('cuda', torch.cuda.current_device())

    # If model exists - continue training (trainer.py:30-39)
    model_path = os.path.join('model', 'model.ckpt')
    if os.path.isfile(model_path):
        model = unet.UNet.load_from_checkpoint(checkpoint_path=model_path)
        model.lr = lr
        model.loss_function = loss_function
    else:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = unet.UNet(lr=lr, num_channels=num_channels, num_classes=num_classes, loss_function=loss_function,
                              architecture=architecture, encoder_name=encoder_name, pretrained=pretrained, dim=dim,
                              act_dtype=act_dtype)
    model = model.to(device)
    rank0 = process_group is None or torch.distributed.get_rank(process_group) == 0
    if process_group is not None:                   # every rank has read the checkpoint before rank 0 removes it
        torch.distributed.barrier(group=process_group)
    if rank0 and os.path.isfile(model_path):        # remove old checkpoint (trainer.py:41-43)
        os.remove(model_path)
    os.makedirs('model', exist_ok=True)
    log_dir = os.path.join('model', 'history', time.strftime('%Y-%m-%d_%H-%M-%S'), 'version_0')
    if rank0:
        os.makedirs(log_dir, exist_ok=True)
    fields = ['epoch', 'step'] + [f'train/{m}' for m in METRICS] + [f'val/{m}' for m in METRICS]
    engine = make_train_engine(model, lr=lr, loss_kind=loss_function.native_kind, process_group=process_group)
    model.train()
    best, step = float('inf'), 0
    for epoch in range(epochs):
        rows = []
        for X, y, w in train_loader:
            rows.append(engine.train_step(X, y, w, sync=False).clone())
            step += 1
        tr = _mean(rows)
        va = _mean([engine.eval_step(X, y, w, sync=False).clone() for X, y, w in val_loader])
        if rank0:
            path = os.path.join(log_dir, 'metrics.csv')
            new = not os.path.isfile(path)
            with open(path, 'a', newline='') as f:
                wr = csv.DictWriter(f, fieldnames=fields)
                if new:
                    wr.writeheader()
                # Lightning's CSVLogger writes the validation row, then the training-epoch row
                wr.writerow({'epoch': epoch, 'step': step - 1, **{f'val/{m}': va[m] for m in METRICS}})
                wr.writerow({'epoch': epoch, 'step': step - 1, **{f'train/{m}': tr[m] for m in METRICS}})
            if va['Loss'] < best:                    # ModelCheckpoint(monitor='val/Loss', mode='min')
                best = va['Loss']
                model.hparams['lr'] = lr
                model.save_checkpoint(model_path)
    return model
