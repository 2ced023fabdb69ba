# -*- coding: utf-8 -*-
"""roctx ranges around the phases of a step (tracing row): with Y4_ROCTX=1 the harness
(yolo/engine/build.py, bench.py) brackets forward / loss / backward / optimizer / postprocess with roctxRangePush / Pop,
so that `rocprofv3 --marker-trace --kernel-trace` groups the kernels of a step by phase.  Off by default: the ranges are
then no-ops and libroctx64.so is never loaded.  (The reference has no tracing hooks; NVTX-style markers are what its
users would add around the same calls, yolo/engine/build.py:60-69,133-137.)"""
import contextlib
import ctypes
import os

_STATE = {'lib': None, 'tried': False}


def enabled():
    return os.environ.get('Y4_ROCTX', '0') == '1'


def _lib():
    if _STATE['tried']:
        return _STATE['lib']
    _STATE['tried'] = True
    for name in ('libroctx64.so', '/opt/rocm/lib/libroctx64.so', 'librocprofiler-sdk-roctx.so'):
        try:
            L = ctypes.CDLL(name)
            L.roctxRangePushA.argtypes = [ctypes.c_char_p]
            L.roctxRangePushA.restype = ctypes.c_int
            L.roctxRangePop.restype = ctypes.c_int
            _STATE['lib'] = L
            break
        except (OSError, AttributeError):
            continue
    return _STATE['lib']


@contextlib.contextmanager
def range(name):
    """`with trace.range('forward'):` -- a roctx range when Y4_ROCTX=1 and the library loads, otherwise nothing."""
    L = _lib() if enabled() else None
    if L is None:
        yield
        return
    L.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        L.roctxRangePop()
