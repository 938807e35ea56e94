"""CPU: the HIP SuperPoint sources, compiled against the test-only emulator, vs the
oracle and the reference's golden vectors on the small golden cases."""
import importlib
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import superpoint_ref
from tests import golden_cases as gc
from tests.parity import compare_superpoint, order_is_reference_like

sp_mod = importlib.import_module("deep-image-matching_amd.superpoint_hip")
GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("name", list(gc.SP_CASES))
def test_superpoint_emulated_vs_golden_and_oracle(emu_lib, name):
    case = gc.SP_CASES[name]
    sd, img = gc.sp_weights(case), gc.sp_image(case)
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=512, device="cpu", lib=emu_lib)
    out = net(img)
    taps = net.debug_taps()
    ref = superpoint_ref.superpoint_forward(img, sd, case["cfg"], taps=True)
    # dense stages
    np.testing.assert_allclose(taps["encoder"][0].permute(2, 0, 1).numpy(), ref["encoder"][0].numpy(), atol=2e-4, rtol=1e-4)
    h, w = ref["logits"].shape[-2:]
    np.testing.assert_allclose(taps["logits"][0].reshape(h, w, 65).permute(2, 0, 1).numpy(), ref["logits"][0].numpy(), atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(taps["score_map"][0].numpy(), ref["score_map"][0].numpy(), atol=2e-6, rtol=1e-4)
    # selection stage must be bit-exact GIVEN the same score map: run the oracle's NMS on OUR map
    nms_on_ours = superpoint_ref.simple_nms(taps["score_map"], case["cfg"]["nms_radius"])
    assert torch.equal(nms_on_ours[0], taps["nms_map"][0])
    yx, sc = superpoint_ref.select_keypoints(taps["nms_map"][0], case["cfg"]["keypoint_threshold"], case["cfg"]["remove_borders"],
                                              case["cfg"]["max_keypoints"])
    assert set(map(tuple, torch.flip(yx, [1]).tolist())) == set(map(tuple, out["keypoints"].long().tolist()))
    # end to end vs the reference's golden outputs
    g = np.load(GOLD / f"sp_{name}.npz")
    gold = {"keypoints": torch.from_numpy(g["keypoints"]), "scores": torch.from_numpy(g["scores"]), "descriptors": torch.from_numpy(g["descriptors"])}
    res = compare_superpoint({k: v.cpu() for k, v in out.items()}, gold)
    k = case["cfg"]["max_keypoints"]
    order_is_reference_like(out, k_limited=(k >= 0 and res["n_out"] == k))


def test_fused_conv1a_equals_the_two_kernel_path(emu_lib):
    """conv1a evaluated inside conv1b's halo staging (conv_x6.hip, F1A) must reproduce the separate
    conv1a kernel + conv1b bit for bit: same fmaf chain, same zero padding on both levels."""
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd = gc.sp_weights(case)
    img = torch.rand(1, 1, 44, 70, generator=torch.Generator().manual_seed(9))  # ragged: partial tiles on both axes
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(44, 70), capacity=512, device="cpu", lib=emu_lib)
    try:
        emu_lib.dim_tune_set(3, 1)
        a = net(img); ta = net.debug_taps()
        emu_lib.dim_tune_set(3, 0)
        b = net(img); tb = net.debug_taps()
    finally:
        emu_lib.dim_tune_set(3, 1)
    # (with pre-split planes the encoder tap is rebuilt from (h + l) / 16, hence the 3e-7; everything downstream is exact)
    np.testing.assert_allclose(ta["encoder"].numpy(), tb["encoder"].numpy(), rtol=3e-7, atol=4e-9)  # small values: the low piece is an fp16 subnormal (absolute step 2^-24 / 16)
    assert torch.equal(ta["score_map"], tb["score_map"]) and torch.equal(ta["logits"], tb["logits"])
    assert torch.equal(a["keypoints"], b["keypoints"]) and torch.equal(a["descriptors"], b["descriptors"])


def test_superpoint_bf16x6_and_fp32_modes_agree_with_the_default(emu_lib):
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd, img = gc.sp_weights(case), gc.sp_image(case)
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=512, device="cpu", lib=emu_lib)
    base = net(img); tb = net.debug_taps()
    try:
        for mode in (1, 0):
            emu_lib.dim_tune_set(1, mode)
            out = net(img); t = net.debug_taps()
            np.testing.assert_allclose(t["encoder"].numpy(), tb["encoder"].numpy(), atol=2e-4, rtol=1e-4)
            assert set(map(tuple, out["keypoints"].long().tolist())) == set(map(tuple, base["keypoints"].long().tolist()))
    finally:
        emu_lib.dim_tune_set(1, 2)


def test_presplit_activation_planes_are_bit_identical_to_consumer_side_splits(emu_lib):
    """fp16x3: conv-to-conv activations stored as pre-split fp16 planes (producer splits once) vs fp32 storage with
    the split in every consumer: the pieces are the same function of the same fp32 value, so nothing may change."""
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd = gc.sp_weights(case)
    img = torch.rand(1, 1, 52, 70, generator=torch.Generator().manual_seed(10))  # ragged tiles at every scale
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(52, 70), capacity=512, device="cpu", lib=emu_lib)
    try:
        emu_lib.dim_tune_set(5, 1)
        a = net(img); ta = net.debug_taps()
        emu_lib.dim_tune_set(5, 0)
        b = net(img); tb = net.debug_taps()
    finally:
        emu_lib.dim_tune_set(5, 1)
    # the encoder tap of the planes path is rebuilt from (h + l) / 16: 22 of fp32's 24 mantissa bits; everything
    # computed FROM the planes is bit-identical
    np.testing.assert_allclose(ta["encoder"].numpy(), tb["encoder"].numpy(), rtol=3e-7, atol=4e-9)  # small values: the low piece is an fp16 subnormal (absolute step 2^-24 / 16)
    assert torch.equal(ta["score_map"], tb["score_map"]) and torch.equal(ta["logits"], tb["logits"])
    assert torch.equal(a["keypoints"], b["keypoints"]) and torch.equal(a["descriptors"], b["descriptors"])


def test_conv_16_row_tiles_are_bit_identical_to_8_row_tiles(emu_lib):
    """conv_x6.hip's 16-row workgroup tile (bit 4 of dim_tune_set key 2, the default for the production shapes) accumulates
    every output in the same order as the 8-row tile: nothing may change."""
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd = gc.sp_weights(case)
    img = torch.rand(1, 1, 52, 70, generator=torch.Generator().manual_seed(11))  # ragged tiles at every scale
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(52, 70), capacity=512, device="cpu", lib=emu_lib)
    try:
        emu_lib.dim_tune_set(2, 1)
        a = net(img); ta = net.debug_taps()
        emu_lib.dim_tune_set(2, 1 | 16)
        b = net(img); tb = net.debug_taps()
    finally:
        emu_lib.dim_tune_set(2, 1 | 16)
    for k in ("encoder", "score_map", "logits"):
        assert torch.equal(ta[k], tb[k]), k
    assert torch.equal(a["keypoints"], b["keypoints"]) and torch.equal(a["descriptors"], b["descriptors"])


def test_lds_dma_staging_is_ordered_both_ways(emu_lib):
    """The emulator lands an LDS-DMA transfer at the issuing thread's s_waitcnt vmcnt(0) by default (a missing wait reads
    stale LDS: RAW).  hipemu_set_dma_mode(1) lands it at ISSUE instead, which exposes the opposite mistake — restaging a
    buffer other threads still read (WAR).  The weight-slice double buffer of conv_x6.hip must be right in both."""
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd, img = gc.sp_weights(case), gc.sp_image(case)
    net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=512, device="cpu", lib=emu_lib)
    a = net(img); ta = net.debug_taps()
    try:
        emu_lib.hipemu_set_dma_mode(1)
        emu_lib.hipemu_set_schedule(1)   # wave after wave between barriers: the waves are maximally out of step
        b = net(img); tb = net.debug_taps()
    finally:
        emu_lib.hipemu_set_dma_mode(0)
        emu_lib.hipemu_set_schedule(1 if os.environ.get("HIPEMU_ORDER") == "wave_serial" else 0)
    for k in ("encoder", "score_map", "logits"):
        assert torch.equal(ta[k], tb[k]), k
    assert torch.equal(a["keypoints"], b["keypoints"]) and torch.equal(a["descriptors"], b["descriptors"])


def test_range_guard_is_silent_on_ragged_odd_sized_batches(emu_lib):
    """Benign images at sizes whose pooled maps have odd pixel counts (the pre-split layout pads every image to a pixel pair) and
    partial tiles on both axes: no kernel may report a value outside the fp16 split's range — the emulator hands out device
    memory poisoned with 3.4e38, so a read of anything that was never written (padding slots, rows past a ragged end) would."""
    capi = importlib.import_module("deep-image-matching_amd.capi")
    name = next(iter(gc.SP_CASES))
    case = gc.SP_CASES[name]
    sd = gc.sp_weights(case)
    for hw in ((45, 70), (52, 38), (77, 102)):
        imgs = torch.rand(2, hw[0], hw[1], generator=torch.Generator().manual_seed(hw[0]))
        net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=2, max_hw=hw, capacity=512, device="cpu", lib=emu_lib)
        capi.saturation(emu_lib, None, reset=True)
        net.extract_batch(imgs)
        total, sites = capi.saturation(emu_lib, None, reset=True)
        assert total == 0, (hw, sites)


@pytest.mark.parametrize("H,W,k", [(24, 2200, -1), (16, 264, 100)])
def test_wide_score_rows_keep_the_reference_candidate_order(emu_lib, H, W, k):
    """count_rows / emit_rows read 16 bytes per lane when the row length allows it: 256-column groups, 1024-column rounds (2200: two rounds, a
    ragged last group).  The candidates must come out in torch.nonzero's row-major order (SPN:183-186) — keypoints identical INCLUDING order."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sp_mod = importlib.import_module("deep-image-matching_amd.superpoint_hip")
    sd = weights.synthetic_superpoint_state_dict(5)
    cfg = {"nms_radius": 2, "keypoint_threshold": 0.0005, "max_keypoints": k, "remove_borders": 2}
    img = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(H + W))
    net = sp_mod.SuperPointHIP(sd, cfg, max_batch=1, max_hw=(H, W), capacity=8192, device="cpu", lib=emu_lib)
    out = net(img)
    ref = superpoint_ref.superpoint_forward(img, sd, cfg)
    assert torch.equal(out["keypoints"].cpu(), ref["keypoints"]) and ref["keypoints"].shape[0] > 50
    assert (out["scores"].cpu() - ref["scores"]).abs().max().item() < 1e-5


@pytest.mark.parametrize("k", [4097, 9000])
def test_top_k_above_4096_keypoints(emu_lib, k):
    """config/superpoint+superglue.yaml:9 asks for max_keypoints 8000: above the one-workgroup kernel's 4096 the selection is a radix select
    into a global key table, 4096-key chunk sorts and a rank merge (sp_post.hip).  Same keypoints in the same order as torch.topk on the same
    NMS map (scores of a noise image are distinct): 2 chunks (4097: one real key in the second) and 3 chunks."""
    case = gc.SP_CASES["noise_topk"]
    sd = gc.sp_weights(case)
    H, W = 160, 224
    img = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(5))
    cfg = {**case["cfg"], "nms_radius": 0, "keypoint_threshold": 0.0, "max_keypoints": k}     # radius 0: every pixel inside the border is a candidate (~32 000)
    net = sp_mod.SuperPointHIP(sd, cfg, max_batch=1, max_hw=(H, W), capacity=k, device="cpu", lib=emu_lib)
    out = net(img)
    taps = net.debug_taps()
    yx, sc = superpoint_ref.select_keypoints(taps["nms_map"][0], cfg["keypoint_threshold"], cfg["remove_borders"], k)
    assert out["keypoints"].shape[0] == k
    assert torch.equal(torch.flip(yx, [1]), out["keypoints"].long()) and torch.equal(sc, out["scores"])


def test_fused_detector_tail_equals_the_two_kernel_path(emu_lib):
    """convPb + 65-way softmax + depth-to-space as one kernel (gemm_x6_head_kernel, dim_tune_set key 16) must give the score map of the
    GEMM + softmax_d2s_kernel path BIT FOR BIT (it is the tap every NMS test pins): same logits, same reduction association, same
    exponentials.  Ragged sizes: the last 128-cell block is partial and cell rows wrap inside a block."""
    case = gc.SP_CASES[next(iter(gc.SP_CASES))]
    sd = gc.sp_weights(case)
    for H, W, B in ((44, 70, 1), (96, 128, 2), (160, 264, 1)):
        img = torch.rand(B, H, W, generator=torch.Generator().manual_seed(H))
        net = sp_mod.SuperPointHIP(sd, case["cfg"], max_batch=B, max_hw=(H, W), capacity=512, device="cpu", lib=emu_lib)
        try:
            emu_lib.dim_tune_set(16, 1)
            a = [t.clone() for t in net.extract_batch(img)]; ta = net.debug_taps(B)
            emu_lib.dim_tune_set(16, 0)
            b = [t.clone() for t in net.extract_batch(img)]; tb = net.debug_taps(B)
        finally:
            emu_lib.dim_tune_set(16, 1)
        assert torch.equal(ta["score_map"], tb["score_map"]), (H, W)
        assert torch.equal(ta["logits"], tb["logits"])            # (the fused path rebuilds the tap with the plain GEMM)
        assert torch.equal(a[3], b[3])
        for i in range(B):   # (slots past the live count are uninitialised)
            k = int(a[3][i])
            assert all(torch.equal(x[i, :k], y[i, :k]) for x, y in zip(a[:3], b[:3]))
