import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle (torch CPU) is the slow half of every parity test; torch defaults to one thread per host CPU — 256 on the GPU box, where the
    # oracle's LightGlue runs 3 x SLOWER on 128+ threads than on 16 (bench.py cpu_baseline: 8 thr 0.25, 16 thr 0.28, 64 thr 0.19, 128 thr 0.08
    # pairs/s).  VERDICT r4 next #9: keep the -m gpu suite well under the driver's limit.
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 16))
    except Exception:
        pass


@pytest.fixture(scope="session")
def emu_lib():
    """The HIP sources compiled against tests/hipemu (CPU fibers). Test-only."""
    import importlib

    build = importlib.import_module("deep-image-matching_amd.build")
    import ctypes

    lib = ctypes.CDLL(str(build.build_emu()))
    lib.dim_last_error.restype = ctypes.c_char_p
    return lib


@pytest.fixture(scope="session")
def emu_research_lib():
    """The emulator build WITH -DDIM_RESEARCH (the default-off prototypes of dim_tune_set keys 14 / 15: their bit-identity tests)."""
    import ctypes
    import importlib

    build = importlib.import_module("deep-image-matching_amd.build")
    lib = ctypes.CDLL(str(build.build_emu(research=True)))
    lib.dim_last_error.restype = ctypes.c_char_p
    return lib


@pytest.fixture(scope="session")
def hip_research_lib():
    """lib/libdim_hip_research.so on the GPU (built by __graft_entry__.build(); travels with the snapshot): the prototype variants' GPU tests."""
    import importlib

    capi = importlib.import_module("deep-image-matching_amd.capi")
    p = capi.LIB_PATH.parent / "libdim_hip_research.so"
    if not p.exists():
        pytest.skip("libdim_hip_research.so not built (python __graft_entry__.py build)")
    return capi.load(str(p))


@pytest.fixture(scope="session")
def hip_lib():
    """The real gfx950 library through the package loader (fails loudly without a GPU)."""
    import importlib

    capi = importlib.import_module("deep-image-matching_amd.capi")
    return capi.load()


@pytest.fixture
def emu_install(emu_lib):
    """Routes the package loader to the emulator build (capi.install test hook) for the duration of one test, so the
    plugin classes are constructed exactly as in production — no library / device arguments."""
    import importlib

    capi = importlib.import_module("deep-image-matching_amd.capi")
    capi.install(emu_lib, "cpu")
    try:
        yield emu_lib
    finally:
        emu_lib.dim_tune_set(1, 2)
        capi.set_arithmetic(emu_lib, 2)
        capi.install(None)
