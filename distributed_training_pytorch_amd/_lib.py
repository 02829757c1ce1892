"""ctypes face of the native libraries (``lib/libdtp_hip.so``, ``lib/libdtp_runtime.so``).

The kernels export a plain C ABI (raw device pointers + ``hipStream_t``), so there is no torch
C++ extension to compile: ``hipcc`` builds the library in seconds and PyTorch-ROCm tensors are
passed by ``data_ptr()`` on the caller's current HIP stream (which also makes every launch
capturable into a hipGraph through ``torch.cuda.graph``).

On a GPU box a missing library is a hard error — there is no silent PyTorch fallback on CUDA
devices.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")
_lock = threading.Lock()
_hip = None
_rt = None

c_void_p, c_int, c_float, c_size_t, c_long, c_ull = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                     ctypes.c_size_t, ctypes.c_long, ctypes.c_ulonglong)

_SIGS = {
    "dtp_igemm_nt": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                     c_float, c_float, c_ull, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                     c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "dtp_wgrad_tt": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                     c_int, c_float, c_int, c_int, c_void_p],
    "dtp_conv_c64": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                     c_float, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "dtp_wgrad_c64": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                      c_float, c_void_p],
    "dtp_maxpool2x2_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dtp_maxpool2x2_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                           ctypes.POINTER(c_int), c_void_p],
    "dtp_quant_fp8": [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p],
    "dtp_colsum_parts": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "dtp_avgpool_flatten_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtp_avgpool_flatten_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtp_colsum": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "dtp_transpose_bf16": [c_void_p, c_void_p, c_int, c_int, c_long, c_long, c_int, c_long, c_long, c_void_p],
    "dtp_transpose_multi": [c_void_p, c_int, c_int, c_void_p],
    "dtp_nchw_to_nhwc_bf16": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                              c_void_p, c_void_p, c_void_p],
    "dtp_conv1_fprop": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "dtp_conv1_wgrad": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                        c_int, c_float, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p],
    "dtp_cast_f32_bf16": [c_void_p, c_void_p, c_size_t, c_void_p],
    "dtp_relu_mask": [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p],
    "dtp_stream_delay": [c_float, c_void_p],
    "dtp_softmax_xent": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_int, c_float, c_float, c_void_p],
    "dtp_bn_fwd_train": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                         c_float, c_float, c_int, c_void_p, c_int, c_void_p],
    "dtp_bn_fwd_eval": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float,
                        c_int, c_void_p],
    "dtp_bn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                   c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p],
    "dtp_bn_fwd_train_pool": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int, c_void_p],
    "dtp_bn_bwd_pool": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                        c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "dtp_augment_u8": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "dtp_sgd_flat": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                     c_float, c_float, c_int, c_int, c_void_p],
    "dtp_adamw_flat": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float,
                       c_float, c_float, c_float, c_float, c_int, c_void_p],
}


def lib_path(name="libdtp_hip.so"):
    # DTP_HIP_LIB: another build of the kernel library, for same-box A/B runs of two source states
    # (device-to-device clock spread on this pool is larger than most kernel changes)
    if name == "libdtp_hip.so" and os.environ.get("DTP_HIP_LIB"):
        return os.path.abspath(os.environ["DTP_HIP_LIB"])
    return os.path.join(_LIBDIR, name)


def _load(name, sigs):
    path = lib_path(name)
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: run `python -m distributed_training_pytorch_amd._build` "
            "(or __graft_entry__.build()) before using CUDA tensors" % path)
    lib = ctypes.CDLL(path)
    for fn, args in sigs.items():
        if not hasattr(lib, fn):
            continue
        f = getattr(lib, fn)
        f.argtypes = args
        f.restype = c_int
    return lib


def hip():
    """The kernel library; raises (loudly) when it has not been built."""
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                _hip = _load("libdtp_hip.so", _SIGS)
    return _hip


def runtime():
    global _rt
    if _rt is None:
        with _lock:
            if _rt is None:
                _rt = ctypes.CDLL(lib_path("libdtp_runtime.so"))
    return _rt


def hip_available():
    return os.path.exists(lib_path("libdtp_hip.so"))


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("native kernel %s failed with code %d" % (what, rc))


class _Workspace(object):
    """Per-device scratch buffers that persist across steps (allocation-free hot path)."""

    def __init__(self):
        self._bufs = {}

    def get(self, name, numel, dtype, device):
        key = (name, dtype, device.index if device.index is not None else torch.cuda.current_device())
        t = self._bufs.get(key)
        if t is None or t.numel() < numel:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    def zero_page(self, device):
        key = ("zero_page", torch.uint8, device.index if device.index is not None else torch.cuda.current_device())
        t = self._bufs.get(key)
        if t is None:
            t = torch.zeros(4096, dtype=torch.uint8, device=device)
            self._bufs[key] = t
        return t


workspace = _Workspace()
