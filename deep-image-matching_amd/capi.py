"""ctypes binding of libdim_hip.so (the C ABI declared in include/dim_hip.h).

There is deliberately no CPU fallback: if the gfx950 library cannot be loaded
or no GPU is visible, loading fails with an exception."""
from __future__ import annotations

import ctypes
from pathlib import Path

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "lib" / "libdim_hip.so"


class DimHipError(RuntimeError):
    pass


def load(path: str | None = None) -> ctypes.CDLL:
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise DimHipError(f"{p} not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950)")
    lib = ctypes.CDLL(str(p))
    lib.dim_last_error.restype = ctypes.c_char_p
    if path is None:
        _LIB = lib
    return lib


def check(lib, rc: int) -> None:
    if rc != 0:
        msg = lib.dim_last_error().decode(errors="replace")
        if "out of memory" in msg.lower():
            # keep the substring the reference's tile fallback keys on (matcher_base.py:251-256)
            raise DimHipError("CUDA out of memory (HIP): " + msg)
        raise DimHipError(msg)


def ptr(t):
    """Device (or, under the test emulator, host) pointer of a torch tensor / None."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
