"""Python face of the native RCCL bucket transport (``csrc/runtime/rccl_transport.cpp``).

The reducer hands each ready bucket to :meth:`launch`; the C++ side records an event on the
compute stream, makes its own (default-priority) communication stream wait for it and issues
``ncclAllReduce(..., ncclAvg)`` there, so communication overlaps with the rest of backward on
separate HIP streams.  :meth:`wait` makes the compute stream wait (device-side, no host sync) for
all collectives before the optimizer runs.  The RCCL unique id is created on rank 0 and shared
through the already-initialised ``torch.distributed`` process group (any backend), which stays
in charge of barriers / object broadcasts.
"""
import atexit
import ctypes
import os
import sys
import threading
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
        rt.dtp_rccl_reduce_scatter_async.argtypes = [vp, vp, st, ci, vp]
        rt.dtp_rccl_reduce_scatter_async.restype = ci
        rt.dtp_rccl_all_gather_async.argtypes = [vp, vp, st, ci, vp, ci]
        rt.dtp_rccl_all_gather_async.restype = ci
        rt.dtp_rccl_comm_stream.argtypes = [vp]
        rt.dtp_rccl_comm_stream.restype = vp
        pvp, pst = ctypes.POINTER(vp), ctypes.POINTER(st)
        rt.dtp_reducer_set_shard.argtypes = [vp, ci, vp, vp, ci, pvp, pvp, pvp, pvp, pvp, pst, pvp, ci, ci]
        rt.dtp_reducer_set_shard.restype = ci
        rt.dtp_reducer_begin_pass.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ci, ctypes.POINTER(ci), ci]
        rt.dtp_reducer_begin_pass.restype = ci
        rt._dtp_typed = True
    return rt


class NativeRcclTransport(object):
    def __init__(self, rank=None, world=None, device=None, unique_id=None):
        if not torch.cuda.is_available():
            raise RuntimeError("NativeRcclTransport needs a GPU")
        self.rt = _rt()
        self.rank = comm.rank() if rank is None else rank
        self.world = comm.world_size() if world is None else world
        self.device = torch.cuda.current_device() if device is None else device
        n = self.rt.dtp_rccl_unique_id_bytes()
        payload = unique_id
        if payload is None:      # stand-alone use (tests, world 1): create / share the id here
            buf = ctypes.create_string_buffer(n)
            if self.rank == 0:
                if self.rt.dtp_rccl_unique_id(buf, n) != n:
                    raise RuntimeError("ncclGetUniqueId failed")
            payload = bytes(buf.raw)
            if self.world > 1:
                payload = comm.broadcast_object(payload, src=0)
        if len(payload) != n:
            raise ValueError("RCCL unique id must be %d bytes" % n)
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

    def launch_sum(self, flat):
        """In-place SUM (no averaging) — the exact fp32 hand-over of sharded mode (reducer.py)."""
        if flat.dtype not in _KIND or not flat.is_cuda or not flat.is_contiguous():
            raise ValueError("RCCL transport reduces contiguous CUDA f32/bf16 buffers")
        rc = self.rt.dtp_rccl_allreduce_async(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                              _KIND[flat.dtype], 0, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_allreduce_async failed: %d" % rc)
        return True

    def wait(self, handle, flat):
        if handle is None:
            return
        rc = self.rt.dtp_rccl_wait(self.handle, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_wait failed: %d" % rc)

    def reduce_scatter(self, flat):
        """In-place average; rank r's slice of ``flat`` receives the result (sharded mode)."""
        if flat.dtype not in _KIND or not flat.is_cuda or not flat.is_contiguous() or flat.numel() % self.world:
            raise ValueError("reduce_scatter: contiguous CUDA f32/bf16 buffer with numel % world == 0")
        rc = self.rt.dtp_rccl_reduce_scatter_async(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                                   _KIND[flat.dtype], self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_reduce_scatter_async failed: %d" % rc)
        return True

    def all_gather(self, flat):
        """In-place: every rank contributes its slice of ``flat``."""
        if flat.dtype not in _KIND or not flat.is_cuda or not flat.is_contiguous() or flat.numel() % self.world:
            raise ValueError("all_gather: contiguous CUDA f32/bf16 buffer with numel % world == 0")
        rc = self.rt.dtp_rccl_all_gather_async(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                               _KIND[flat.dtype], self._stream(), 0)
        if rc != 0:
            raise RuntimeError("dtp_rccl_all_gather_async failed: %d" % rc)
        return True

    def broadcast(self, flat, src=0):
        if self.world == 1:
            return
        rc = self.rt.dtp_rccl_broadcast(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                        _KIND[flat.dtype], src, self._stream())
        if rc != 0:
            raise RuntimeError("dtp_rccl_broadcast failed: %d" % rc)

    def make_scheduler(self, bucket_tensors, param_bucket_lists, shard=None):
        """Native bucket scheduler (C++): ``bucket_tensors[b]`` is the flat gradient range of bucket
        b, ``param_bucket_lists[i]`` the bucket indices parameter i contributes to; ``shard`` (see
        ``BucketedReducer._native_shard_spec``) switches it to reduce-scatter -> optimizer ->
        all-gather per bucket."""
        return NativeBucketScheduler(self, bucket_tensors, param_bucket_lists, shard)

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

    def __init__(self, transport, bucket_tensors, param_bucket_lists, shard=None):
        self.transport = transport
        self.opt_kind = 0
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
        if shard is not None:
            self._set_shard(shard)

    def _set_shard(self, spec):
        """Hand the C++ scheduler this rank's slices of (master, gradient, optimizer state, working
        weights) per bucket and the fused optimizer kernel it runs between the reduce-scatter and
        the all-gather."""
        from ..ops.optim import FusedAdamW, FusedSGD
        opt, a, w = spec["optimizer"], spec["arena"], spec["weights"]
        rank, world = spec["rank"], spec["world"]
        hip = _lib.hip()
        if isinstance(opt, FusedSGD):
            self.opt_kind, st = 1, [opt._flat_state["momentum_buffer"], None]
        elif isinstance(opt, FusedAdamW):
            self.opt_kind, st = 2, [opt._flat_state["exp_avg"], opt._flat_state["exp_avg_sq"]]
        else:
            raise ValueError("native sharded mode supports FusedSGD / FusedAdamW")
        if w.dtype not in _KIND:
            raise ValueError("working weights must be f32 or bf16")
        nb = len(spec["ranges"])
        vp = ctypes.c_void_p

        def arr(t):
            if t is None:
                return None
            es = t.element_size()
            return (vp * nb)(*[t.data_ptr() + (lo + rank * ((hi - lo) // world)) * es for lo, hi in spec["ranges"]])
        counts = (ctypes.c_size_t * nb)(*[(hi - lo) // world for lo, hi in spec["ranges"]])
        wbase = (vp * nb)(*[w.data_ptr() + lo * w.element_size() for lo, hi in spec["ranges"]])
        self._keep += [a.flat_master, a.flat_grad, w] + [t for t in st if t is not None]
        rc = self.rt.dtp_reducer_set_shard(
            self.handle, self.opt_kind, ctypes.cast(hip.dtp_sgd_flat, vp), ctypes.cast(hip.dtp_adamw_flat, vp), nb,
            arr(a.flat_master), arr(a.flat_grad), arr(st[0]), arr(st[1]), arr(w), counts, wbase, _KIND[w.dtype], world)
        if rc != 0:
            raise RuntimeError("dtp_reducer_set_shard failed: %d" % rc)

    def begin_pass(self, opt, consts):
        """Per backward pass: the optimizer's scalar arguments (read when the pass starts, so LR
        schedulers and the first-step flag are honoured)."""
        g = opt.param_groups[0]
        if self.opt_kind == 1:
            hyper = [g["lr"], g["momentum"], g["dampening"], g["weight_decay"]]
            iargs = [int(bool(g["nesterov"])), int(bool(consts["first"]))]
        else:
            b1, b2 = g["betas"]
            hyper = [g["lr"], b1, b2, g["eps"], g["weight_decay"]]
            iargs = [int(consts["step"])]
        h = (ctypes.c_float * len(hyper))(*[float(x) for x in hyper])
        ia = (ctypes.c_int * len(iargs))(*iargs)
        rc = self.rt.dtp_reducer_begin_pass(self.handle, h, len(hyper), ia, len(iargs))
        if rc != 0:
            raise RuntimeError("dtp_reducer_begin_pass failed: %d" % rc)

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


class _Deadline(object):
    """Passive bring-up deadline.  The timer thread touches neither the GPU nor the process group:
    when the deadline passes it only reports where the main thread is stuck and ends the process
    with ``launch.EXIT_NATIVE_BRINGUP`` — a fresh set of workers (the self-launcher relaunches with
    ``DTP_TRANSPORT=torch``) is the recovery, never a half-built communicator next to a live job."""

    def __init__(self, seconds, what):
        self.what = what
        self.timer = threading.Timer(seconds, self._expire)
        self.timer.daemon = True
        self.seconds = seconds

    def _expire(self):
        from ..launch import EXIT_NATIVE_BRINGUP
        sys.stderr.write("[dtp] rank %d: %s did not finish within %.0f s (DTP_NATIVE_INIT_TIMEOUT_S); "
                         "exiting %d so the launcher can restart the job with DTP_TRANSPORT=torch\n"
                         % (comm.rank(), self.what, self.seconds, EXIT_NATIVE_BRINGUP))
        try:
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:
            pass
        sys.stderr.flush()
        os._exit(EXIT_NATIVE_BRINGUP)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def bring_up(device, want="auto"):
    """Create the native transport on every rank — on the CALLING (main) thread — and agree on the
    outcome.  Returns ``(transport_or_None, info)``; ``None`` means every rank uses the process group.

    Order of events (every collective on the process group happens on this thread, in the same
    order on every rank; nothing runs in a helper thread):
      1. each rank loads ``libdtp_runtime.so`` and resolves its symbols; rank 0 creates the RCCL
         unique id; the id is broadcast and the load results are min-reduced through the process
         group.  With ``backend="nccl"`` that reduction runs on the GPU, i.e. it also proves that
         RCCL can build a communicator between exactly these processes (the process group's own),
         under the process group's timeout;
      2. blocking ``ncclCommInitRank`` for the gradient communicator + one self-checked all-reduce
         on the caller's current stream, under a passive deadline (see :class:`_Deadline`);
      3. the per-rank verdicts are min-reduced; unless all ranks succeeded, the ranks that did
         close their communicator and everybody falls back (``auto``) or raises (``native``).
    """
    info = dict(requested=want, kind="torch", fell_back=False, reason=None)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    # Deadline for step 2 (ncclCommInitRank + probe): 120 s under ANY launcher.  Step 1 has already
    # built a communicator between exactly these processes (the process group's own, on the nccl
    # backend), so a healthy second one comes up in seconds; 120 s is ~20x the slowest bring-up seen.
    # The trade: under the self-launcher (it exports DTP_RESTART_COUNT) an expired deadline costs one
    # relaunch on the process-group transport; under a foreign launcher (torchrun: the driver's N > 1
    # runs) nobody relaunches and the job ends with exit code 75 — but it ends after 2 minutes with a
    # message and a stack dump, inside the caller's own time budget (the driver allows a bench run
    # 600 s), instead of sitting in ncclCommInitRank until that budget kills it without a word
    # (round 2 waited 480 s here).  Callers that prefer the process-group transport outright set
    # DTP_TRANSPORT=torch; a longer wait is DTP_NATIVE_INIT_TIMEOUT_S.
    default_timeout = "120"
    timeout_s = float(os.environ.get("DTP_NATIVE_INIT_TIMEOUT_S", default_timeout))
    ok, payload, n, err = 1.0, b"", 0, None
    try:
        rt = _rt()
        n = rt.dtp_rccl_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        if comm.rank() == 0 and rt.dtp_rccl_unique_id(buf, n) != n:
            raise RuntimeError("ncclGetUniqueId failed")
        payload = bytes(buf.raw)
    except Exception as e:      # library missing / symbol mismatch / RCCL refused: a clean, agreed fallback
        ok, err = 0.0, "%s: %s" % (type(e).__name__, e)
    payload = comm.broadcast_object(payload, src=0)
    agreed = comm.all_reduce_scalars([ok if payload else 0.0], op="min")[0]
    native = None
    if agreed >= 1.0:
        ok = 0.0
        with _Deadline(timeout_s, "native RCCL bring-up (ncclCommInitRank + probe all-reduce)"):
            try:
                native = NativeRcclTransport(device=dev_index, unique_id=payload)
                probe = torch.full((256,), float(comm.rank() + 1), device=device)
                native.wait(native.launch(probe), probe)
                torch.cuda.current_stream().synchronize()
                expect = (comm.world_size() + 1) / 2.0        # mean of 1..world
                ok = 1.0 if bool(((probe - expect).abs() < 1e-5).all()) else 0.0
                if ok < 1.0:
                    err = "probe all-reduce returned %r, expected %r" % (float(probe[0]), expect)
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, e)
        agreed = comm.all_reduce_scalars([ok], op="min")[0]
    if agreed >= 1.0:
        info.update(kind="native")
        return native, info
    if native is not None:
        native.close()
    info.update(fell_back=True, reason=err or "another rank failed")
    if want == "native":
        raise RuntimeError("DTP_TRANSPORT=native but the RCCL transport did not come up on every rank (%s)"
                           % info["reason"])
    if comm.rank() == 0:
        import warnings
        warnings.warn("native RCCL transport unavailable (%s); gradients go through the torch.distributed "
                      "process group" % info["reason"])
    return None, info
