"""Python face of the native RCCL bucket transport (``csrc/runtime/rccl_transport.cpp``).

The reducer hands each ready bucket to :meth:`launch`; the C++ side records an event on the
compute stream, makes its own high-priority communication stream wait for it and issues
``ncclAllReduce(..., ncclAvg)`` there, so communication overlaps with the rest of backward on
separate HIP streams.  :meth:`wait` makes the compute stream wait (device-side, no host sync) for
all collectives before the optimizer runs.  The RCCL unique id is created on rank 0 and shared
through the already-initialised ``torch.distributed`` process group (any backend), which stays
in charge of barriers / object broadcasts.
"""
import atexit
import ctypes
import weakref

import torch

from .. import _lib
from . import comm

_KIND = {torch.float32: 0, torch.bfloat16: 1}
_live = weakref.WeakSet()


@atexit.register
def _close_all():
    """Destroy communicators while the HIP runtime is still up (before interpreter teardown)."""
    for t in list(_live):
        try:
            t.close()
        except Exception:
            pass


def _rt():
    rt = _lib.runtime()
    if not getattr(rt, "_dtp_typed", False):
        vp, ci, st = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        rt.dtp_rccl_unique_id.argtypes = [vp, ci]
        rt.dtp_rccl_unique_id.restype = ci
        rt.dtp_rccl_unique_id_bytes.restype = ci
        rt.dtp_rccl_create.argtypes = [vp, ci, ci, ci]
        rt.dtp_rccl_create.restype = vp
        rt.dtp_rccl_allreduce_async.argtypes = [vp, vp, st, ci, ci, vp]
        rt.dtp_rccl_allreduce_async.restype = ci
        rt.dtp_rccl_wait.argtypes = [vp, vp]
        rt.dtp_rccl_wait.restype = ci
        rt.dtp_rccl_broadcast.argtypes = [vp, vp, st, ci, ci, vp]
        rt.dtp_rccl_broadcast.restype = ci
        rt.dtp_rccl_sync.argtypes = [vp]
        rt.dtp_rccl_sync.restype = ci
        rt.dtp_rccl_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
        rt.dtp_rccl_stats.restype = ci
        rt.dtp_rccl_destroy.argtypes = [vp]
        rt.dtp_rccl_destroy.restype = ci
        ull_p = ctypes.POINTER(ctypes.c_ulonglong)
        rt.dtp_reducer_create.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(st), ctypes.POINTER(ci), ci,
                                          ctypes.POINTER(ci), ctypes.POINTER(ci)]
        rt.dtp_reducer_create.restype = vp
        rt.dtp_reducer_mark_ready.argtypes = [vp, ci, vp]
        rt.dtp_reducer_mark_ready.restype = ci
        rt.dtp_reducer_finalize.argtypes = [vp, vp]
        rt.dtp_reducer_finalize.restype = ci
        rt.dtp_reducer_set_sync.argtypes = [vp, ci]
        rt.dtp_reducer_set_sync.restype = ci
        rt.dtp_reducer_stats.argtypes = [vp, ull_p, ull_p, ull_p]
        rt.dtp_reducer_stats.restype = ci
        rt.dtp_reducer_destroy.argtypes = [vp]
        rt.dtp_reducer_destroy.restype = ci
        rt._dtp_typed = True
    return rt


class NativeRcclTransport(object):
    def __init__(self, rank=None, world=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("NativeRcclTransport needs a GPU")
        self.rt = _rt()
        self.rank = comm.rank() if rank is None else rank
        self.world = comm.world_size() if world is None else world
        self.device = torch.cuda.current_device() if device is None else device
        n = self.rt.dtp_rccl_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        if self.rank == 0:
            if self.rt.dtp_rccl_unique_id(buf, n) != n:
                raise RuntimeError("ncclGetUniqueId failed")
        payload = bytes(buf.raw)
        if self.world > 1:
            payload = comm.broadcast_object(payload, src=0)
        idbuf = ctypes.create_string_buffer(payload, n)
        self.handle = self.rt.dtp_rccl_create(idbuf, self.rank, self.world, self.device)
        if not self.handle:
            raise RuntimeError("ncclCommInitRank failed (rank %d of %d)" % (self.rank, self.world))
        _live.add(self)

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(self, flat):
        if flat.dtype not in _KIND or not flat.is_cuda or not flat.is_contiguous():
            raise ValueError("RCCL transport reduces contiguous CUDA f32/bf16 buffers")
        rc = self.rt.dtp_rccl_allreduce_async(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                              _KIND[flat.dtype], 1, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_allreduce_async failed: %d" % rc)
        return True

    def wait(self, handle, flat):
        if handle is None:
            return
        rc = self.rt.dtp_rccl_wait(self.handle, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_wait failed: %d" % rc)

    def broadcast(self, flat, src=0):
        if self.world == 1:
            return
        rc = self.rt.dtp_rccl_broadcast(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                        _KIND[flat.dtype], src, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_broadcast failed: %d" % rc)

    def make_scheduler(self, bucket_tensors, param_bucket_lists):
        """Native bucket scheduler (C++): ``bucket_tensors[b]`` is the flat gradient range of bucket
        b, ``param_bucket_lists[i]`` the bucket indices parameter i contributes to."""
        return NativeBucketScheduler(self, bucket_tensors, param_bucket_lists)

    def stats(self):
        a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        self.rt.dtp_rccl_stats(self.handle, ctypes.byref(a), ctypes.byref(b))
        return dict(launched=a.value, bytes=b.value)

    def close(self):
        if getattr(self, "handle", None):
            self.rt.dtp_rccl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeBucketScheduler(object):
    """Python face of ``dtp_reducer_*`` (csrc/runtime/rccl_transport.cpp): ready-counting, launch
    of a bucket's all-reduce the moment its last gradient is queued, and the end-of-backward
    hand-back of the comm stream all happen in C++ — one ctypes call per notification."""

    def __init__(self, transport, bucket_tensors, param_bucket_lists):
        self.transport = transport
        self.rt = transport.rt
        self._keep = list(bucket_tensors)                # the ranges must outlive the scheduler
        nb, npar = len(bucket_tensors), len(param_bucket_lists)
        for t in bucket_tensors:
            if t.dtype not in _KIND or not t.is_cuda or not t.is_contiguous():
                raise ValueError("bucket ranges must be contiguous CUDA f32/bf16 tensors")
        ptrs = (ctypes.c_void_p * nb)(*[t.data_ptr() for t in bucket_tensors])
        counts = (ctypes.c_size_t * nb)(*[t.numel() for t in bucket_tensors])
        kinds = (ctypes.c_int * nb)(*[_KIND[t.dtype] for t in bucket_tensors])
        offs, idx = [0], []
        for lst in param_bucket_lists:
            idx.extend(int(b) for b in lst)
            offs.append(len(idx))
        c_offs = (ctypes.c_int * (npar + 1))(*offs)
        c_idx = (ctypes.c_int * max(1, len(idx)))(*(idx or [0]))
        self.handle = self.rt.dtp_reducer_create(transport.handle, nb, ptrs, counts, kinds, npar, c_offs, c_idx)
        if not self.handle:
            raise RuntimeError("dtp_reducer_create rejected the bucket plan")

    def mark_ready(self, param_index):
        n = self.rt.dtp_reducer_mark_ready(self.handle, int(param_index), NativeRcclTransport._stream())
        if n < 0:
            raise RuntimeError("dtp_reducer_mark_ready failed: %d" % n)
        return n

    def finalize(self):
        n = self.rt.dtp_reducer_finalize(self.handle, NativeRcclTransport._stream())
        if n < 0:
            raise RuntimeError("dtp_reducer_finalize failed: %d" % n)
        return n

    def set_sync(self, on):
        self.rt.dtp_reducer_set_sync(self.handle, 1 if on else 0)

    def stats(self):
        a, b, c = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        self.rt.dtp_reducer_stats(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return dict(passes=a.value, launched=b.value, duplicate_notifications=c.value)

    def close(self):
        if getattr(self, "handle", None):
            self.rt.dtp_reducer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
