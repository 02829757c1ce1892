"""Throughput / step-time meters (the reference has none: SURVEY.md §5.1, §5.5 — only tqdm it/s).

``ThroughputMeter`` measures items/sec over a window bracketed by device synchronisation;
``StepTimer`` records per-phase HIP-event timings (data wait / forward / backward / optimizer)
without forcing a host sync per step: events are resolved lazily at ``summary()``.
"""
import time

import torch


class ThroughputMeter(object):
    def __init__(self):
        self.items = 0
        self.elapsed = 0.0
        self._t0 = None

    def start(self):
        self.items = 0
        self._t0 = time.perf_counter()

    def add(self, n):
        self.items += int(n)

    def stop(self, sync_device=None):
        if sync_device is not None and torch.device(sync_device).type == "cuda":
            torch.cuda.synchronize(sync_device)
        self.elapsed = time.perf_counter() - (self._t0 if self._t0 is not None else time.perf_counter())
        return self.items / self.elapsed if self.elapsed > 0 else 0.0


class StepTimer(object):
    def __init__(self, enabled=True):
        self.enabled = enabled and torch.cuda.is_available()
        self._open = {}
        self._pairs = {}

    def begin(self, name):
        if self.enabled:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._open[name] = ev

    def end(self, name):
        if self.enabled and name in self._open:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._pairs.setdefault(name, []).append((self._open.pop(name), ev))

    def reset(self):
        self._open, self._pairs = {}, {}

    def phase(self, name):
        """``with timer.phase("fwd"): ...`` — a pair of HIP events around the block (no host sync)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            self.begin(name)
            try:
                yield
            finally:
                self.end(name)
        return _cm()

    def summary(self):
        if not self.enabled:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self._pairs.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = dict(n=len(ms), mean_ms=sum(ms) / max(len(ms), 1), total_ms=sum(ms))
        return out
