"""roctx ranges for rocprofv3: ``with trace_range("fwd"): ...`` marks a span that
shows up in ``rocprofv3 --marker-trace``; a no-op when the roctx library is absent or
``DTP_ROCTX`` is not set, so the hot path pays nothing by default.
"""
import contextlib
import ctypes
import os

_lib = None
_tried = False


def _roctx():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("DTP_ROCTX", "0") != "1":
        return None
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _lib = ctypes.CDLL(name)
            _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            return _lib
        except OSError:
            continue
    return None


@contextlib.contextmanager
def trace_range(name):
    lib = _roctx()
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib is not None:
            lib.roctxRangePop()
