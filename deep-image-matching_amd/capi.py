"""ctypes binding of libdim_hip.so (the C ABI declared in include/dim_hip.h).

There is deliberately no CPU fallback: if the gfx950 library cannot be loaded
or no GPU is visible, loading fails with an exception."""
from __future__ import annotations

import ctypes
from pathlib import Path

_LIB = None
_INSTALLED_DEVICE = None
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


def install(lib, device: str = "cpu") -> None:
    """TEST HOOK (tests/ only): make ``load()`` return an already opened library object — the CPU tests install the
    emulator build of the same sources here, with ``device`` = "cpu" — instead of injecting it through the plugin
    constructors.  ``install(None)`` restores the product behaviour.  The package itself never calls this."""
    global _LIB, _INSTALLED_DEVICE
    _LIB, _INSTALLED_DEVICE = lib, (device if lib is not None else None)
    if lib is not None:
        lib.dim_last_error.restype = ctypes.c_char_p


def installed_device():
    """Device forced by ``install`` (None in the product: the plugins then use their own device)."""
    return _INSTALLED_DEVICE


class LgRawFeatures(ctypes.Structure):
    """include/dim_hip.h: dim_lg_raw_features (one image's keypoints / descriptors as uploaded, for dim_lg_stage_features)."""
    _fields_ = [("kpts_dev", ctypes.c_void_p), ("desc_dev", ctypes.c_void_p), ("n", ctypes.c_int), ("kpts_f16", ctypes.c_int),
                ("desc_f16", ctypes.c_int), ("desc_is_dn", ctypes.c_int)]


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


# ---- arithmetic selection and the fp16x3 range guard ------------------------------------------------------
ARITHMETIC = {"fp16x3": 2, "bf16x6": 1, "fp32": 0}
SAT_SITES = 16  # DIM_SAT_SITES
SAT_NAMES = ["sp_image", "sp_encoder", "sp_heads", "lg_input", "lg_qkv", "lg_ffn", "lg_desc", "op", "aliked"]
_arith = {}  # id(lib) -> current mode (the library default is fp16x3)


def get_arithmetic(lib) -> int:
    return _arith.get(id(lib), 2)


def set_arithmetic(lib, mode) -> int:
    """Process-wide DEFAULT matrix arithmetic (dim_tune_set key 1); returns the previous mode.  Handles may override it for themselves
    (set_handle_arithmetic)."""
    mode = ARITHMETIC[mode] if isinstance(mode, str) else int(mode)
    prev = get_arithmetic(lib)
    lib.dim_tune_set(1, mode)
    _arith[id(lib)] = mode
    return prev


def set_handle_arithmetic(lib, handle, mode) -> None:
    """Matrix arithmetic of ONE extractor / matcher handle (dim_handle_tune_set key 1); ``None`` = back to the process default."""
    v = -1 if mode is None else (ARITHMETIC[mode] if isinstance(mode, str) else int(mode))
    check(lib, lib.dim_handle_tune_set(handle, 1, v))


def saturation(lib, stream=None, reset: bool = True):
    """(total, {site: count}) of the fp16x3 range-guard counters (dim_saturation_read; synchronises `stream`)."""
    counts = (ctypes.c_uint * SAT_SITES)()
    total = ctypes.c_ulonglong()
    check(lib, lib.dim_saturation_read(counts, ctypes.byref(total), int(reset), stream))
    named = {SAT_NAMES[i] if i < len(SAT_NAMES) else str(i): int(counts[i]) for i in range(SAT_SITES) if counts[i]}
    return int(total.value), named


class SaturationError(DimHipError):
    pass


def run_guarded(lib, stream, fn, what: str, policy: str = "fallback", logger=None, handle=None, arithmetic=None):
    """Runs ``fn()`` (one extract / match call enqueued on ``stream``) under the fp16x3 range guard: when the
    default arithmetic is active and a kernel reported a value outside the exact range of the fp16 split
    (|x| > 4094), the call is repeated in bf16x6 (no range limit; policy "fallback") or a SaturationError is
    raised (policy "raise").  Synchronises the stream.  Other arithmetic modes run unguarded.
    ``handle`` / ``arithmetic``: the library handle ``fn`` calls and its own arithmetic override (None = process default) — the re-run then
    switches THAT handle to bf16x6 (dim_handle_tune_set) instead of the process-wide default, so other handles / threads are not affected."""
    mode = get_arithmetic(lib) if arithmetic is None else (ARITHMETIC[arithmetic] if isinstance(arithmetic, str) else int(arithmetic))
    if mode != 2 or policy == "off":
        return fn()
    check(lib, lib.dim_saturation_reset(stream))  # drop anything a previous unguarded call left behind (one enqueued memset: no read-back, no synchronisation)
    out = fn()
    total, sites = saturation(lib, stream, reset=True)
    if total == 0:
        return out
    if policy == "raise":
        raise SaturationError(f"{what}: fp16x3 range exceeded at {sites}; set arithmetic='bf16x6'")
    if logger is not None:
        logger.warning("%s: fp16x3 range exceeded at %s - repeating the call in bf16x6", what, sites)
    if handle is not None:
        set_handle_arithmetic(lib, handle, 1)
        try:
            return fn()
        finally:
            set_handle_arithmetic(lib, handle, arithmetic)
    prev = set_arithmetic(lib, 1)
    try:
        return fn()
    finally:
        set_arithmetic(lib, prev)
