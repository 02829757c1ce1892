"""Dropout random-number state as part of the RUN state.

Parity target: the reference draws dropout masks from each process's own torch generator
(``model/vgg16.py:40,73,76``): every rank sees an independent stream, and nothing about it is saved
(``trainer/trainer.py:85-93``), so a resumed run is not reproducible.

Here a dropout mask is a pure function of ``(seed, rank, forward-call number, layer, element)``:
the kernels hash ``key + element index`` with a counter-based generator (``rng_uniform`` in
``csrc/dtp_common.h``), and the key is

    key(layer) = mix(seed, rank) + calls * STEP + salt(layer)            (mod 2**64)

* ``seed``  — ``Trainer(seed=)`` / ``torch.initial_seed()``;
* ``rank``  — the data-parallel rank: two ranks never draw the same mask;
* ``calls`` — training forwards of the model so far (micro-steps under gradient accumulation);
  it is the ONLY mutable part, travels in the snapshot (``rng_state``) and makes
  ``train 4 steps == train 2 + snapshot + resume + 2`` hold bit for bit;
* ``salt``  — position of the layer inside its model (stable across processes and re-creations).

The hipGraph step keeps ``mix(seed, rank) + calls * STEP`` in device memory and advances it with a
captured add (``engine/graph.py``); the eager path computes the same number on the host, so both
draw identical masks.
"""
import os

import torch

MASK64 = (1 << 64) - 1
STEP = 0x5851F42D4C957F2D & ((1 << 62) - 1)
SALT = 0x9E3779B97F4A7C15


def _mix64(a, b):
    z = (a * 0x9E3779B97F4A7C15 + b * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def _current_rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return int(os.environ.get("RANK", "0") or 0)


class DropoutRng(object):
    def __init__(self, seed=None, rank=None):
        self.seed = seed           # None: torch.initial_seed() at first use
        self.rank = rank           # None: the process's data-parallel rank at first use
        self.calls = 0

    def _resolved(self):
        if self.seed is None:
            self.seed = int(torch.initial_seed())
        if self.rank is None:
            self.rank = _current_rank()
        return int(self.seed), int(self.rank)

    def base(self):
        """``mix(seed, rank)``, kept below 2**62 so that the device-side int64 copy never overflows
        into the sign bit over any realistic number of steps."""
        seed, rank = self._resolved()
        return _mix64(seed & MASK64, rank) & ((1 << 62) - 1)

    def advance(self):
        self.calls += 1

    def device_value(self):
        """What the hipGraph step keeps in device memory: ``base + calls * STEP`` (as a signed 64-bit
        integer; the kernels add in unsigned 64-bit arithmetic, wrap-around is part of the function)."""
        v = (self.base() + self.calls * STEP) & MASK64
        return v - (1 << 64) if v >= (1 << 63) else v

    def key(self, salt):
        return (self.base() + self.calls * STEP + salt) & MASK64

    @staticmethod
    def salt_for(index):
        return ((index + 1) * SALT) & ((1 << 62) - 1)

    # ------------------------------------------------------------------ snapshot
    def state_dict(self):
        seed, _ = self._resolved()
        return dict(seed=seed, dropout_calls=int(self.calls))

    def load_state_dict(self, state):
        if not isinstance(state, dict):
            return
        if state.get("seed") is not None:
            self.seed = int(state["seed"])
        self.calls = int(state.get("dropout_calls", 0))
