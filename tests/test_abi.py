"""CPU: the gfx950 library loads and exports every symbol include/dim_hip.h declares (no compute)."""
import ctypes
import importlib
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "dim_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(dim_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_abi():
    syms = _declared_symbols()
    for s in ("dim_sp_create", "dim_sp_extract", "dim_sp_destroy", "dim_lg_create", "dim_lg_match", "dim_lg_destroy",
              "dim_last_error", "dim_profile_start", "dim_profile_stop"):
        assert s in syms


def test_hip_library_builds_and_exports_every_declared_symbol():
    build = importlib.import_module("deep-image-matching_amd.build")
    lib = ctypes.CDLL(str(build.build_hip()))  # hipcc cross-compiles gfx950 without a GPU
    for s in _declared_symbols():
        assert hasattr(lib, s), f"libdim_hip.so does not export {s}"
    lib.dim_abi_version.restype = ctypes.c_int
    assert lib.dim_abi_version() == 2


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the product wrappers must refuse to run (loudly), not fall back."""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sp = importlib.import_module("deep-image-matching_amd.superpoint_hip")
    capi = importlib.import_module("deep-image-matching_amd.capi")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    with pytest.raises(capi.DimHipError):
        sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(0), {}, device="cpu")
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    with pytest.raises(RuntimeError):
        plugins.SuperPointExtractor({"general": {}, "extractor": {}})


def test_product_package_never_imports_the_oracle():
    for p in (ROOT / "deep-image-matching_amd").glob("*.py"):
        t = p.read_text()
        assert "import oracle" not in t and "from oracle" not in t, p


def test_product_library_rejects_the_research_keys_and_the_research_build_has_them():
    """dim_tune_set keys 12-15 (timing probes that give wrong results by design, prototypes that lost their A/B, the Winograd conv1b) exist only in
    libdim_hip_research.so (build.build_research); the product library returns an error for them and compiles none of those kernels."""
    build = importlib.import_module("deep-image-matching_amd.build")
    lib = ctypes.CDLL(str(build.build_hip()))
    lib.dim_last_error.restype = ctypes.c_char_p
    for key in (12, 13, 14, 15):
        assert lib.dim_tune_set(key, 1) != 0 and b"research" in lib.dim_last_error()
    assert lib.dim_tune_set(19, 0) != 0 and lib.dim_tune_set(-1, 0) != 0
    assert lib.dim_tune_set(1, 2) == 0 and lib.dim_tune_set(11, 3) == 0
    assert not hasattr(lib, "dim_conv_wg_phase_read")
    text = (ROOT / "include" / "dim_hip.h").read_text()
    assert "wrong results" not in text                        # the product header documents no wrong-results mode
    rlib = ctypes.CDLL(str(build.build_research()))
    for key, v in ((12, 0), (13, 0), (14, 32), (15, 0)):
        assert rlib.dim_tune_set(key, v) == 0
    assert hasattr(rlib, "dim_conv_wg_phase_read")
    # the prototype / probe kernels are not in the product binary at all (their names appear in the embedded code object's symbol table)
    prod, res = (build.LIBDIR / "libdim_hip.so").read_bytes(), (build.LIBDIR / "libdim_hip_research.so").read_bytes()
    for name in (b"conv3x3_wg_f1a_kernel", b"gemm_x6_probe_kernel", b"gemm_x6_qkv_kc64_kernel", b"gemm_x6_ffn_fused_step_kernel", b"gemm_x6_stream_kernel"):
        assert name in res and name not in prod, name
