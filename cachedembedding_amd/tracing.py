"""Named ranges around the phases of the training loop, as the reference brackets them with
``torch.profiler.record_function`` ("prefetch cache", forward, backward, optimisation: recsys/dlrm_main.py:258-282).

`phase(name)` opens a roctx range (visible in ``rocprofv3 --marker-trace`` timelines next to the kernels) when
libroctx64 is present, and a torch profiler range; it costs two C calls and is a no-op for the GPU."""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

_roctx = None
for _p in (os.environ.get("ROCTX_LIB"), "libroctx64.so", "/opt/rocm/lib/libroctx64.so"):
    if not _p:
        continue
    try:
        _roctx = ctypes.CDLL(_p)
        _roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _roctx.roctxRangePushA.restype = ctypes.c_int
        _roctx.roctxRangePop.restype = ctypes.c_int
        break
    except (OSError, AttributeError):
        _roctx = None


_NAMES: dict = {}


@contextlib.contextmanager
def phase(name: str):
    # (the torch range only while a torch profiler is collecting: record_function costs microseconds per use even
    # when nobody listens, and this brackets every cache op of a prefetch_num = 1 loop)
    if _roctx is not None:
        b = _NAMES.get(name)
        if b is None:
            b = _NAMES[name] = name.encode()
        _roctx.roctxRangePushA(b)
    try:
        if torch.autograd._profiler_enabled():
            with torch.profiler.record_function(name):
                yield
        else:
            yield
    finally:
        if _roctx is not None:
            _roctx.roctxRangePop()
