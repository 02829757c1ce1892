"""Snapshot I/O in the reference's schema.

Parity target: reference ``trainer/trainer.py:85-101`` — one ``torch.save`` file per snapshot with
the keys ``epoch``, ``model_state_dict`` (unwrapped module, fp32), ``optimizer_state_dict``,
``scheduler_state_dict`` under ``<save_folder>/weights/<name>.pth``.

Additions that old readers ignore: ``best_fitness`` (quirk #7), ``rng_state``,
``framework``.  Writes are atomic (tmp + ``os.replace``).  Loading uses ``weights_only=True``
first (nothing in our own files needs unpickling arbitrary objects) and maps to CPU; the caller
moves state to the training device (fixes quirk #4).
"""
import os

import torch

REQUIRED_KEYS = ("epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict")


def save_snapshot(path, epoch, model, optimizer, scheduler, extra=None):
    snapshot = dict(epoch=int(epoch),
                    model_state_dict=model.state_dict(),
                    optimizer_state_dict=optimizer.state_dict() if optimizer is not None else {},
                    scheduler_state_dict=scheduler.state_dict() if scheduler is not None else {})
    if extra:
        for k, v in extra.items():
            if k not in snapshot:
                snapshot[k] = v
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(snapshot, tmp)
    os.replace(tmp, path)
    return path


def load_snapshot(path):
    try:
        snap = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        # our own files may carry python scalars / Counters inside scheduler state
        snap = torch.load(path, map_location="cpu", weights_only=False)
    for k in REQUIRED_KEYS:
        if k not in snap:
            raise KeyError("snapshot %s lacks key %r" % (path, k))
    return snap
