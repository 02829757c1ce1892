"""Snapshot I/O in the reference's schema.

Parity target: reference ``trainer/trainer.py:85-101`` — one ``torch.save`` file per snapshot with
the keys ``epoch``, ``model_state_dict`` (unwrapped module, fp32), ``optimizer_state_dict``,
``scheduler_state_dict`` under ``<save_folder>/weights/<name>.pth``.

Additions that old readers ignore: ``best_fitness`` (quirk #7), ``global_step``,
``rng_state`` (written by ``Trainer._rng_state``: the model's dropout seed + call counter of
``ops/rng.py`` and torch's CPU generator state — a resumed run draws the masks the uninterrupted
run would have drawn) and ``framework``.  Writes are atomic (tmp + ``os.replace``).  Loading ALWAYS uses
``weights_only=True`` — a snapshot is data, never code: the one non-tensor type our files carry
(``collections.Counter`` for ``MultiStepLR.milestones``) is allow-listed explicitly, and a file the
safe loader refuses is an error unless the caller opts in with ``trusted=True`` (a file they wrote
themselves with an older torch).  State maps to CPU; the caller moves it to the training device
(fixes quirk #4).
"""
import collections
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


def load_snapshot(path, trusted=False):
    """Load a snapshot without executing anything from the file.  ``trusted=True`` (explicit, for
    files you wrote yourself) allows full unpickling."""
    if trusted:
        snap = torch.load(path, map_location="cpu", weights_only=False)
    else:
        with torch.serialization.safe_globals([collections.Counter]):
            snap = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(snap, dict):
        raise TypeError("snapshot %s does not hold a dict" % path)
    for k in REQUIRED_KEYS:
        if k not in snap:
            raise KeyError("snapshot %s lacks key %r" % (path, k))
    return snap
