"""CPU (emulator): the fp16x3 range guard and the per-output-channel weight scales on adversarial inputs
(tests/adversarial.py).  A call that leaves the exact range of the fp16 split must (a) bump the device counters and
(b) come back correct through the bf16x6 re-run of the guarded entry points; everything inside the range must be
fp32-accurate WITHOUT a re-run."""
import importlib

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref
from tests import adversarial as adv
from tests.parity import compare_lightglue, compare_superpoint

capi = importlib.import_module("deep-image-matching_amd.capi")
sp_mod = importlib.import_module("deep-image-matching_amd.superpoint_hip")
lg_mod = importlib.import_module("deep-image-matching_amd.lightglue_hip")

CFG = {"nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 60, "remove_borders": 2}


@pytest.mark.parametrize("name", adv.SP_CASES)
def test_superpoint_adversarial_ranges(emu_lib, name):
    sd, img, expect_guard = adv.sp_case(name, 40, 56)
    net = sp_mod.SuperPointHIP(sd, CFG, max_batch=1, max_hw=(40, 56), capacity=256, device="cpu", lib=emu_lib)
    capi.saturation(emu_lib, None, reset=True)
    net.extract_batch(img[0].contiguous())            # raw call in the default arithmetic
    total, sites = capi.saturation(emu_lib, None, reset=True)
    assert (total > 0) == expect_guard, (name, sites)
    if name == "bright":
        assert "sp_image" in sites
    out = net(img)                                      # guarded call: re-runs in bf16x6 when the guard fired
    assert capi.get_arithmetic(emu_lib) == 2            # the fallback restores the default arithmetic
    taps = net.debug_taps()
    ref = superpoint_ref.superpoint_forward(img, sd, CFG, taps=True)
    assert (taps["score_map"][0] - ref["score_map"][0]).abs().max().item() <= 2e-5, name
    compare_superpoint({k: v.cpu() for k, v in out.items()}, ref)
    with pytest.raises(capi.SaturationError) if expect_guard else _noraise():
        strict = sp_mod.SuperPointHIP(sd, CFG, max_batch=1, max_hw=(40, 56), capacity=256, device="cpu", lib=emu_lib, on_saturation="raise")
        strict(img)


# (the input-side cases desc_1e5 / tiny_desc go through the same init kernel on both paths: small-batch kernels only)
@pytest.mark.parametrize("name,big", [(n, b) for n in adv.LG_CASES for b in (False, True) if not (b and n in ("desc_1e5", "tiny_desc"))])
def test_lightglue_adversarial_ranges(emu_lib, name, big):
    """big: the kernels that only large batches select — 128 x 256 GEMM blocks writing the K | V tile images (dim_tune_set 6 = 2) and
    the one-kernel feed-forward (11 = 4), whose range guards sit in different code — forced at this size."""
    sd, f0, f1, conf, expect_guard = adv.lg_case(name, m=40, n=36, n_layers=2)
    if big:
        emu_lib.dim_tune_set(6, 2); emu_lib.dim_tune_set(11, 4)
    try:
        _lightglue_adversarial(emu_lib, name, sd, f0, f1, conf, expect_guard)
    finally:
        emu_lib.dim_tune_set(6, 1); emu_lib.dim_tune_set(11, 3)


def _lightglue_adversarial(emu_lib, name, sd, f0, f1, conf, expect_guard):
    net = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=64, device="cpu", lib=emu_lib)
    data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
    net.on_saturation = "off"
    capi.saturation(emu_lib, None, reset=True)
    net(data)
    total, sites = capi.saturation(emu_lib, None, reset=True)
    assert (total > 0) == expect_guard, (name, sites)
    net.on_saturation = "fallback"
    res = net(data, dense=True)
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
    la = ref["log_assignment"]
    m, n = la.shape[0] - 1, la.shape[1] - 1
    # 1e-3 absolute on the dense log-assignment, or (descriptors x 1e5: entries of order 1e9) the same relative to its range
    tol = 1e-3 * max(1.0, la[:m, :n].abs().max().item() * 1e-3)
    info = compare_lightglue(res, ref, score_tol=1e-3, dense_ref=la, dense_out=res["dense"], dense_tol=tol)
    assert info["max_log_assignment_diff"] <= tol


class _noraise:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_weight_scales_are_per_output_channel(emu_lib):
    """One 1e4 x outlier COLUMN must not cost the other columns any accuracy (dim_x3_create scales per column)."""
    import ctypes

    g = torch.Generator().manual_seed(0)
    K, N, M = 64, 128, 64
    A = torch.randn(M, K, generator=g)
    Wt = torch.randn(K, N, generator=g) * 0.05
    Wt[:, 7] *= 1e4
    Wt[:, 9] *= 1e-4
    h, npad = ctypes.c_void_p(), ctypes.c_int()
    assert emu_lib.dim_x3_create(ctypes.c_void_p(Wt.contiguous().data_ptr()), K, N, ctypes.byref(h), ctypes.byref(npad)) == 0
    C = torch.zeros(M, N)
    assert emu_lib.dim_op_gemm_x6_f32(ctypes.c_void_p(A.data_ptr()), K, h, npad.value, None, None, 0, ctypes.c_void_p(C.data_ptr()), N, M, N, K, 0, None) == 0
    emu_lib.dim_x3_destroy(h)
    ref = (A.double() @ Wt.double())
    scale = (A.double().abs() @ Wt.double().abs())
    rel = ((C.double() - ref).abs() / scale).max().item()
    assert rel < 4e-7, rel


def test_aliked_range_guard_and_the_three_arithmetics(emu_lib):
    """ALIKED's full- / half-resolution convolutions and GEMMs run as fp16x3 on the matrix cores (aliked_x3.hip, gemm_x6.hip) under
    the default arithmetic: (a) a benign image leaves the guard silent and equals the oracle; (b) the SAME network through the
    fp32 VALU / fp32-MFMA paths (dim_tune_set(1, 0)) gives the same keypoints — the two implementations check each other;
    (c) an image 5000x out of range trips DIM_SAT_ALIKED and the guarded call comes back correct through the fp32 paths."""
    from oracle import aliked_ref
    from tests import golden_cases as gc
    from tests.test_aliked_emu import compare_aliked
    al_mod = importlib.import_module("deep-image-matching_amd.aliked_hip")
    case = gc.AL_CASES["rgb_pad"]
    sd, img = gc.al_weights(case), gc.al_image(case)
    net = al_mod.AlikedHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=4096, device="cpu", lib=emu_lib)
    hwc = img[0].permute(1, 2, 0).contiguous()[None]
    capi.saturation(emu_lib, None, reset=True)
    kp, sc, de, n = net.extract_batch(hwc)
    total, sites = capi.saturation(emu_lib, None, reset=True)
    assert total == 0, sites
    ref = aliked_ref.aliked_forward(img, sd, case["cfg"])
    k = int(n[0])
    compare_aliked({"keypoints": kp[0, :k], "scores": sc[0, :k], "descriptors": de[0, :k].t()}, ref)
    prev = capi.set_arithmetic(emu_lib, 0)
    try:
        kp0, sc0, de0, n0 = net.extract_batch(hwc)
    finally:
        capi.set_arithmetic(emu_lib, prev)
    k0 = int(n0[0])
    compare_aliked({"keypoints": kp0[0, :k0], "scores": sc0[0, :k0], "descriptors": de0[0, :k0].t()}, ref)
    assert k0 == k and (de0[0, :k] - de[0, :k]).abs().max().item() < 1e-4 and not torch.equal(de0[0, :k], de[0, :k])   # two different code paths
    big = img * 5000.0                                  # |image| > 4094: outside the exact range of the fp16 split
    net.extract_batch(big[0].permute(1, 2, 0).contiguous()[None])
    total, sites = capi.saturation(emu_lib, None, reset=True)
    assert total > 0 and "aliked" in sites
    out = {k_: v.cpu() for k_, v in net(big).items()}   # guarded: repeated on the fp32 paths
    assert capi.get_arithmetic(emu_lib) == 2
    # train-mode BatchNorm makes the network (nearly) invariant to the input scale: same oracle call on the scaled image
    compare_aliked(out, aliked_ref.aliked_forward(big, sd, case["cfg"]))


def test_phase_guards_follow_the_handle_arithmetic(emu_lib):
    """the plugin option ``arithmetic`` is a per-handle override; the phase guards of the batched drivers (pipeline._guarded, pairs.py,
    async_export.py) must decide from THAT handle's mode and re-run THAT handle in bf16x6.  Process default bf16x6 + a handle set to fp16x3 on an
    image that leaves the fp16 range: before the fix the guard read the process default, ran the phase unguarded and returned the saturated
    result."""
    pipeline = importlib.import_module("deep-image-matching_amd.pipeline")
    sd, img, expect_guard = adv.sp_case("bright", 40, 56)
    assert expect_guard
    ref = superpoint_ref.superpoint_forward(img, sd, CFG)
    prev = capi.set_arithmetic(emu_lib, "bf16x6")
    try:
        net = sp_mod.SuperPointHIP(sd, CFG, max_batch=1, max_hw=(40, 56), capacity=256, device="cpu", lib=emu_lib, arithmetic="fp16x3")
        capi.saturation(emu_lib, None, reset=True)
        net.extract_batch(img[0].contiguous())
        assert capi.saturation(emu_lib, None, reset=True)[0] > 0            # the handle really runs fp16x3 under a bf16x6 process default
        kp, sc, de, n = pipeline._guarded(net, lambda: net.extract_batch(img[0].contiguous()), "phase")
        k = int(n[0])
        compare_superpoint({"keypoints": kp[0, :k], "scores": sc[0, :k], "descriptors": de[0, :k].t()}, ref)
        assert capi.get_arithmetic(emu_lib) == 1                             # the process default was not touched
        capi.saturation(emu_lib, None, reset=True)
        net.extract_batch(img[0].contiguous())
        assert capi.saturation(emu_lib, None, reset=True)[0] > 0            # and the handle is back on its own fp16x3 after the re-run
    finally:
        capi.set_arithmetic(emu_lib, prev)
