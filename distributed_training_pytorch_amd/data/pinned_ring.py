"""Python face of the native pinned host ring (``csrc/runtime/pinned_ring.cpp``): page-locked host
slots allocated once, host->device copies on the ring's own HIP copy stream, device-side event
hand-over to the compute stream.

Parity target: the reference pins every batch in its DataLoader (``main.py:16`` ``pin_memory=True``)
and copies it with a blocking ``.to("cuda:k")`` inside the step (``example_trainer.py:70``); here
nothing is pinned per batch and the host never waits for a copy.
"""
import ctypes

import torch

from .. import _lib


def _rt():
    rt = _lib.runtime()
    if not getattr(rt, "_dtp_ring_typed", False):
        vp, ci, st, ull = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_ulonglong
        rt.dtp_ring_create.argtypes = [ci, st, ci]
        rt.dtp_ring_create.restype = vp
        rt.dtp_ring_slot.argtypes = [vp, ci]
        rt.dtp_ring_slot.restype = vp
        rt.dtp_ring_slot_bytes.argtypes = [vp]
        rt.dtp_ring_slot_bytes.restype = st
        rt.dtp_ring_h2d.argtypes = [vp, ci, st, vp, st, vp, ci]
        rt.dtp_ring_h2d.restype = ci
        rt.dtp_ring_consumer_wait.argtypes = [vp, ci, vp]
        rt.dtp_ring_consumer_wait.restype = ci
        rt.dtp_ring_slot_busy.argtypes = [vp, ci]
        rt.dtp_ring_slot_busy.restype = ci
        rt.dtp_ring_host_wait.argtypes = [vp, ci]
        rt.dtp_ring_host_wait.restype = ci
        rt.dtp_ring_fill_u8.argtypes = [vp, ci, ci, st, st, ci, ull, ci]
        rt.dtp_ring_fill_u8.restype = ci
        rt.dtp_ring_stats.argtypes = [vp, ctypes.POINTER(ull), ctypes.POINTER(ull)]
        rt.dtp_ring_stats.restype = ci
        rt.dtp_ring_copy_stream.argtypes = [vp]
        rt.dtp_ring_copy_stream.restype = vp
        rt.dtp_ring_destroy.argtypes = [vp]
        rt.dtp_ring_destroy.restype = ci
        rt._dtp_ring_typed = True
    return rt


class PinnedRing(object):
    def __init__(self, n_slots, slot_bytes, device):
        self.rt = _rt()
        self.device = torch.device(device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.n_slots, self.slot_bytes = int(n_slots), int(slot_bytes)
        self.handle = self.rt.dtp_ring_create(self.n_slots, self.slot_bytes, self.index)
        if not self.handle:
            raise RuntimeError("pinned ring: could not allocate %d x %d bytes of page-locked memory"
                               % (self.n_slots, self.slot_bytes))
        self._views = []
        for i in range(self.n_slots):
            ptr = self.rt.dtp_ring_slot(self.handle, i)
            buf = (ctypes.c_ubyte * self.slot_bytes).from_address(ptr)
            self._views.append(torch.frombuffer(buf, dtype=torch.uint8))      # shares the pinned memory
        # torch's view of the native copy stream: allocations made under it belong to its pool
        self.stream = torch.cuda.ExternalStream(self.rt.dtp_ring_copy_stream(self.handle), device=self.device)

    def host_view(self, slot, offset=0, nbytes=None, dtype=torch.uint8, shape=None):
        """CPU tensor aliasing slot memory [offset, offset + nbytes)."""
        nbytes = self.slot_bytes - offset if nbytes is None else nbytes
        v = self._views[slot][offset:offset + nbytes]
        if dtype != torch.uint8:
            v = v.view(dtype)
        return v.view(shape) if shape is not None else v

    def h2d(self, slot, offset, dst, consumer_stream=None, last=True):
        """Queue slot[offset : offset + dst.nbytes] -> ``dst`` (a CUDA tensor) on the copy stream.
        ``consumer_stream``: the copy additionally waits for everything queued there so far (the
        previous reader of ``dst``); None = ``dst`` is a fresh allocation nobody reads yet."""
        if not dst.is_cuda or not dst.is_contiguous():
            raise ValueError("h2d destination must be a contiguous CUDA tensor")
        nbytes = dst.numel() * dst.element_size()
        cs = consumer_stream if consumer_stream is not None else self.stream
        rc = self.rt.dtp_ring_h2d(self.handle, slot, offset, ctypes.c_void_p(dst.data_ptr()), nbytes,
                                  ctypes.c_void_p(cs.cuda_stream), int(bool(last)))
        if rc != 0:
            raise RuntimeError("dtp_ring_h2d failed: %d" % rc)

    def consumer_wait(self, slot, stream=None):
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        rc = self.rt.dtp_ring_consumer_wait(self.handle, slot, ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("dtp_ring_consumer_wait failed: %d" % rc)

    def slot_busy(self, slot):
        rc = self.rt.dtp_ring_slot_busy(self.handle, slot)
        if rc < 0:
            raise RuntimeError("dtp_ring_slot_busy failed: %d" % rc)
        return rc == 1

    def host_wait(self, slot):
        rc = self.rt.dtp_ring_host_wait(self.handle, slot)
        if rc != 0:
            raise RuntimeError("dtp_ring_host_wait failed: %d" % rc)

    def fill_u8(self, slot, count, image_bytes, label_off, num_classes, seed, threads=8):
        rc = self.rt.dtp_ring_fill_u8(self.handle, slot, count, image_bytes, label_off, num_classes, seed, threads)
        if rc != 0:
            raise RuntimeError("dtp_ring_fill_u8 failed: %d" % rc)

    def stats(self):
        a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        self.rt.dtp_ring_stats(self.handle, ctypes.byref(a), ctypes.byref(b))
        return dict(copies=a.value, bytes=b.value)

    def close(self):
        if getattr(self, "handle", None):
            self._views = []
            self.rt.dtp_ring_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
