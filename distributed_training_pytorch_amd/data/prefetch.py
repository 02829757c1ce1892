"""Asynchronous host->device input pipeline.

The reference copies each batch with a blocking ``tensor.to("cuda:k")`` inside the training step
(``example_trainer.py:70``) although the loader pins memory
(``main.py:16``).  Here:

* ``DevicePrefetcher`` wraps any loader: batch i+1 is copied on a dedicated HIP copy stream while
  batch i computes, and an event hands the tensors over to the compute stream (no host wait).
  Batches that arrive unpinned are staged through a native ring of page-locked slots that is
  allocated ONCE (``data/pinned_ring.py``, ``csrc/runtime/pinned_ring.cpp``) — never
  ``pin_memory()`` per batch; already pinned batches (``DataLoader(pin_memory=True)``) are copied
  straight from where they are.  On CPU it is a transparent pass-through.
* ``PinnedRingSource`` is the benchmark's host-side input: uint8 NCHW batches produced into the
  pinned ring by native threads, one H2D copy per step on the copy stream INSIDE the timed region,
  normalisation fused into the first convolution (``bench.py --data host-u8``).
"""
import time

import torch

from ..utils.profiler import trace_range


def _align(n, a=256):
    return (n + a - 1) // a * a


class StreamStallMeter(object):
    """How long a stream sat in a ``wait_event`` — the GPU-side definition of "starved for input".
    An event recorded right before the wait completes when the stream's EARLIER work is done, one
    recorded right after it completes when the awaited copy has landed as well: the difference is
    time the compute stream had nothing to run because its batch was not there.  (Host-side time in
    ``next(loader)`` cannot tell this from back-pressure: a host that runs ahead of the GPU blocks in
    the loader for most of every step without the GPU ever waiting.)  Events are read lazily, only
    those that completed; never a host synchronisation."""

    def __init__(self):
        self._pairs, self._ms = [], 0.0

    def before(self, stream):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        return e0

    def after(self, e0, stream):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream)
        self._pairs.append((e0, e1))

    def total_ms(self):
        keep = []
        for e0, e1 in self._pairs:
            if e1.query():
                self._ms += e0.elapsed_time(e1)
            else:
                keep.append((e0, e1))
        self._pairs = keep
        return self._ms


class DevicePrefetcher(object):
    def __init__(self, loader, device, depth=2, ring_slots=4):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.ring = None
        self.ring_slots = max(self.depth + 2, int(ring_slots))
        self._k = 0
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.host_wait_s = 0.0          # time the host spent waiting for a ring slot to drain
        self.stall = StreamStallMeter() if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _ensure_ring(self, nbytes):
        if self.ring is None or self.ring.slot_bytes < nbytes:
            from .pinned_ring import PinnedRing
            if self.ring is not None:
                torch.cuda.synchronize(self.device)
                self.ring.close()
            self.ring = PinnedRing(self.ring_slots, _align(int(nbytes * 1.05) + 4096, 4096), self.device)
            self.stream = self.ring.stream
        return self.ring

    def _to_device(self, batch):
        if not self.cuda:
            return batch, None
        tensors = [e for e in batch if torch.is_tensor(e) and e.device.type == "cpu"]
        staged = [e for e in tensors if not e.is_pinned()]
        with trace_range("h2d"):
            slot, offs = None, {}
            if staged:
                # unpinned host tensors: one memcpy into a ring slot (page-locked once, at start-up)
                need = sum(_align(e.numel() * e.element_size()) for e in staged)
                ring = self._ensure_ring(need)
                slot = self._k % ring.n_slots
                self._k += 1
                if ring.slot_busy(slot):
                    t0 = time.perf_counter()
                    ring.host_wait(slot)
                    self.host_wait_s += time.perf_counter() - t0
                off = 0
                for e in staged:
                    nb = e.numel() * e.element_size()
                    ring.host_view(slot, off, nb, e.dtype, tuple(e.shape)).copy_(e.contiguous())
                    offs[id(e)] = off
                    off += _align(nb)
            out = []
            with torch.cuda.stream(self.stream):
                n_staged = len(staged)
                for e in batch:
                    if torch.is_tensor(e) and e.device.type == "cpu":
                        if id(e) in offs:
                            dst = torch.empty(e.shape, dtype=e.dtype, device=self.device)
                            n_staged -= 1
                            self.ring.h2d(slot, offs[id(e)], dst, consumer_stream=None, last=n_staged == 0)
                            out.append(dst)
                        else:
                            out.append(e.to(self.device, non_blocking=True))
                    elif torch.is_tensor(e):
                        out.append(e.to(self.device, non_blocking=True))
                    else:
                        out.append(e)
                ev = torch.cuda.Event()
                ev.record(self.stream)
        return out, ev

    def __iter__(self):
        if getattr(self.loader, "yields_device_batches", False):
            # the loader runs its own copy pipeline (RingSyntheticLoader): pulling ahead of the
            # consumer would recycle its device double buffer under the step that still reads it
            for batch in self.loader:
                yield batch
            return
        it = iter(self.loader)
        queue = []
        try:
            for _ in range(self.depth):
                queue.append(self._to_device(next(it)))
        except StopIteration:
            pass
        while queue:
            batch, ev = queue.pop(0)
            try:
                queue.append(self._to_device(next(it)))
            except StopIteration:
                pass
            if ev is not None:
                cs = torch.cuda.current_stream(self.device)
                e0 = self.stall.before(cs)
                cs.wait_event(ev)
                self.stall.after(e0, cs)
                for e in batch:
                    if torch.is_tensor(e):
                        e.record_stream(torch.cuda.current_stream(self.device))
            yield batch


    def starved_ms_total(self):
        """GPU-side input stall so far (ms): the compute stream waiting for a batch's copy."""
        src = getattr(self.loader, "src", None)
        if getattr(self.loader, "yields_device_batches", False) and hasattr(src, "starved_ms_total"):
            return src.starved_ms_total()
        return self.stall.total_ms() if self.stall is not None else 0.0


class PinnedRingSource(object):
    """Endless iterator of (uint8 NCHW image batch, int64 labels) on the device, fed from the
    pinned host ring.  Synthetic batches are produced into page-locked slots by native threads;
    every ``next()`` hands out the batch whose copy was queued one step earlier and queues the
    following one — slot -> device double buffer on the copy stream, after the event of the
    buffer's previous reader — so one H2D copy per step runs under the step's compute.

    ``refill=False``: ``n_slots`` distinct batches produced once and cycled (the benchmark's
    ``--data host-u8``: the copy is in the timed region, generation is not).  ``refill=True``: a
    producer thread regenerates every slot as soon as its copy has left the host — fresh data each
    step, like a real loader.  ``wait_ms_total()`` = time the host was blocked on the producer."""

    def __init__(self, batch_size, num_classes=3, height=224, width=224, channels=3, device="cuda", seed=0,
                 n_slots=4, threads=8, refill=False):
        from .pinned_ring import PinnedRing
        self.device = torch.device(device)
        self.batch_size, self.num_classes = int(batch_size), int(num_classes)
        self.shape = (self.batch_size, channels, height, width)
        self.image_bytes = channels * height * width
        self.label_off = _align(self.batch_size * self.image_bytes)
        slot_bytes = self.label_off + _align(self.batch_size * 8)
        self.ring = PinnedRing(n_slots, slot_bytes, self.device)
        self.seed, self.threads, self.refill = int(seed), int(threads), bool(refill)
        self.n = self.ring.n_slots
        for p in range(self.n):
            self._produce(p)
        self._ready_upto = self.n - 1          # productions 0 .. n-1 are in their slots
        self.dev = [(torch.empty(self.shape, dtype=torch.uint8, device=self.device),
                     torch.empty(self.batch_size, dtype=torch.int64, device=self.device)) for _ in range(2)]
        # the generator's pixels are ~N(128, 37) + a class band: (x/64 - 2) keeps activations O(1)
        self.normalization = (1.0 / 64.0, (2.0,) * channels, (1.0,) * channels)
        self._wait_s = 0.0
        self.stall = StreamStallMeter()
        self._k = 0
        self._thread = None
        if self.refill:
            import queue
            import threading
            self._cv = threading.Condition()
            self._jobs = queue.Queue()
            self._thread = threading.Thread(target=self._producer, name="dtp-ring-producer", daemon=True)
            self._thread.start()
        self._queue(0)

    def _produce(self, p):
        self.ring.fill_u8(p % self.ring.n_slots, self.batch_size, self.image_bytes, self.label_off, self.num_classes,
                          self.seed * 1000003 + p, self.threads)

    def _producer(self):
        while True:
            p = self._jobs.get()
            if p is None:
                return
            self.ring.host_wait(p % self.n)      # the slot's previous batch has left the host (GIL released)
            self._produce(p)
            with self._cv:
                self._ready_upto = p
                self._cv.notify_all()

    def _queue(self, k):
        if self.refill and k > self._ready_upto:       # production k not in its slot yet: a real data stall
            t0 = time.perf_counter()
            with self._cv:
                while k > self._ready_upto:
                    self._cv.wait(1.0)
            self._wait_s += time.perf_counter() - t0
        slot = k % self.n
        x, y = self.dev[k % 2]
        cs = torch.cuda.current_stream(self.device)
        with trace_range("h2d"):
            self.ring.h2d(slot, 0, x, consumer_stream=cs, last=False)
            self.ring.h2d(slot, self.label_off, y, consumer_stream=cs, last=True)

    def __iter__(self):
        return self

    def __next__(self):
        k = self._k
        self._k += 1
        cs = torch.cuda.current_stream(self.device)
        e0 = self.stall.before(cs)
        self.ring.consumer_wait(k % self.n)              # device-side: compute waits for copy k
        self.stall.after(e0, cs)
        # (without refill the slots are read-only after start-up: nothing on the host ever waits —
        # the host may run several steps ahead of the device, like with device-resident data)
        self._queue(k + 1)
        if self.refill and k >= 1:
            self._jobs.put(k - 1 + self.n)               # slot of batch k-1: regenerate once its copy is done
        return self.dev[k % 2]

    def wait_ms_total(self):
        return 1e3 * self._wait_s

    def starved_ms_total(self):
        """GPU-side input stall so far (ms): the compute stream waiting for its batch's H2D copy."""
        return self.stall.total_ms()

    def close(self):
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join(timeout=30)
            self._thread = None
        torch.cuda.synchronize(self.device)
        self.ring.close()


class RingSyntheticLoader(object):
    """DataLoader-shaped face of :class:`PinnedRingSource` for the Trainer: ``len()`` steps per
    epoch of fresh uint8 batches that are generated by native threads into the pinned ring and
    arrive on the device through the copy stream (``main.py --synthetic --ring-loader``).  Each rank
    generates its own stream (seeded by rank), so there is nothing to shard."""

    yields_device_batches = True      # DevicePrefetcher passes these through untouched

    class _Sampler(object):
        def set_epoch(self, epoch):
            self.epoch = epoch

    def __init__(self, num_samples, batch_size, num_classes, height, width, device, seed=0, threads=8):
        self.steps = max(1, num_samples // batch_size)
        self.sampler = self._Sampler()
        self.src = PinnedRingSource(batch_size, num_classes, height, width, device=device, seed=seed, n_slots=4,
                                    threads=threads, refill=True)
        self.normalization = self.src.normalization

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            x, y = next(self.src)
            yield [x, y]

    def close(self):
        self.src.close()
