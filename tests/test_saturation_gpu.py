"""GPU (MI355X): the fp16x3 range guard, the bf16x6 fallback and the per-output-channel weight scales on the adversarial
inputs of tests/adversarial.py, through the C ABI, vs the oracle.  Also: all three arithmetic modes
at full size, and a yardstick for the dense log-assignment error (reference fp32 vs fp64 next to HIP vs fp64)."""
import importlib

import pytest
import torch

from oracle import lightglue_ref, superpoint_ref
from tests import adversarial as adv
from tests.parity import compare_lightglue, compare_superpoint

pytestmark = pytest.mark.gpu
CFG = {"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": 1000, "remove_borders": 4}


def _mods():
    m = importlib.import_module
    return m("deep-image-matching_amd.capi"), m("deep-image-matching_amd.superpoint_hip"), m("deep-image-matching_amd.lightglue_hip")


@pytest.mark.parametrize("name", adv.SP_CASES)
def test_superpoint_gpu_adversarial_ranges(hip_lib, name):
    capi, sp_mod, _ = _mods()
    H, W = 240, 328   # not multiples of 8 / of the conv tile
    sd, img, expect_guard = adv.sp_case(name, H, W)
    net = sp_mod.SuperPointHIP(sd, CFG, max_batch=1, max_hw=(H, W), capacity=1000)
    stream = net._stream()
    capi.saturation(hip_lib, stream, reset=True)
    net.extract_batch(img[0].contiguous().cuda())
    total, sites = capi.saturation(hip_lib, stream, reset=True)
    assert (total > 0) == expect_guard, (name, sites)
    out = {k: v.cpu() for k, v in net(img.cuda()).items()}      # guarded: bf16x6 re-run when the guard fired
    assert capi.get_arithmetic(hip_lib) == 2
    taps = net.debug_taps()
    ref = superpoint_ref.superpoint_forward(img, sd, CFG, taps=True)
    assert (taps["score_map"][0] - ref["score_map"][0]).abs().max().item() <= 2e-5, name
    assert torch.equal(superpoint_ref.simple_nms(taps["score_map"], CFG["nms_radius"])[0], taps["nms_map"][0])
    compare_superpoint(out, ref)


# (the input-side cases desc_1e5 / tiny_desc go through the same init kernel on both paths: small-batch kernels only)
@pytest.mark.parametrize("name,big", [(n, b) for n in adv.LG_CASES for b in (False, True) if not (b and n in ("desc_1e5", "tiny_desc"))])
def test_lightglue_gpu_adversarial_ranges(hip_lib, name, big):
    """big: the large-batch-only kernels (K | V images from the 128 x 256 projection blocks, dim_tune_set 6 = 2; the one-kernel
    feed-forward, 11 = 4) forced at this size: their range guards are different code."""
    if big:
        hip_lib.dim_tune_set(6, 2); hip_lib.dim_tune_set(11, 4)
    try:
        _lightglue_gpu_adversarial(hip_lib, name)
    finally:
        hip_lib.dim_tune_set(6, 1); hip_lib.dim_tune_set(11, 3)


def _lightglue_gpu_adversarial(hip_lib, name):
    capi, _, lg_mod = _mods()
    sd, f0, f1, conf, expect_guard = adv.lg_case(name, m=500, n=430, n_layers=3)
    net = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=512)
    data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
    net.on_saturation = "off"
    capi.saturation(hip_lib, net._stream(), reset=True)
    net(data)
    total, sites = capi.saturation(hip_lib, net._stream(), reset=True)
    assert (total > 0) == expect_guard, (name, sites)
    net.on_saturation = "fallback"
    res = net(data, dense=True)
    res = {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in res.items()}
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
    la = ref["log_assignment"]
    m, n = la.shape[0] - 1, la.shape[1] - 1
    tol = 1e-3 * max(1.0, la[:m, :n].abs().max().item() * 1e-3)
    compare_lightglue(res, ref, dense_ref=la, dense_out=res["dense"], dense_tol=tol)


@pytest.mark.parametrize("arith", ["fp16x3", "bf16x6", "fp32"])
def test_full_size_pair_in_every_arithmetic_mode(hip_lib, arith):
    """BASELINE configs[2] sizes (1024^2 image, 2048 x 2048 keypoints, 9 layers) in each of the three selectable
    arithmetic modes, vs the oracle: keypoint sets / matches / stop / prune exact, floats within the stated tolerances."""
    capi, sp_mod, lg_mod = _mods()
    weights = importlib.import_module("deep-image-matching_amd.weights")
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0, "pruning_min_kpts": -1}
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(1234), weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    prev = capi.set_arithmetic(hip_lib, arith)
    try:
        ext = sp_mod.SuperPointHIP(sp_sd, cfg, max_batch=1, max_hw=(1024, 1024), capacity=2048)
        mat = lg_mod.LightGlueHIP(lg_sd, conf, max_pairs=1, max_kpts=2048)
        feats, refs = [], []
        for s in (10, 11):
            img = torch.rand(1, 1, 1024, 1024, generator=torch.Generator().manual_seed(s))
            out = {k: v.cpu() for k, v in ext(img.cuda()).items()}
            ref = superpoint_ref.superpoint_forward(img, sp_sd, cfg)
            compare_superpoint(out, ref)
            feats.append(out)
        size = torch.tensor([1024.0, 1024.0])
        data = {f"image{i}": {"keypoints": feats[i]["keypoints"][None], "descriptors": feats[i]["descriptors"].t()[None].contiguous(),
                              "image_size": size[None]} for i in range(2)}
        res = mat(data, dense=True)
        res = {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in res.items()}
        ref = lightglue_ref.lightglue_forward(feats[0]["keypoints"], feats[0]["descriptors"].t().contiguous(), size,
                                              feats[1]["keypoints"], feats[1]["descriptors"].t().contiguous(), size, lg_sd, conf, taps=True)
        info = compare_lightglue(res, ref, dense_ref=ref.get("log_assignment"), dense_out=res["dense"])
        print(arith, info)
    finally:
        capi.set_arithmetic(hip_lib, prev)


def test_log_assignment_error_yardstick(hip_lib):
    """max |delta log-assignment| at 2048 x 2048 of (a) the reference-equivalent fp32 CPU path vs an
    fp64 evaluation of the same network and (b) the HIP path vs the same fp64 evaluation.  The HIP path must be no worse
    than 2x what the reference is against itself (and inside the 1e-3 budget)."""
    capi, _, lg_mod = _mods()
    weights = importlib.import_module("deep-image-matching_amd.weights")
    from tests import golden_cases as gc

    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0, "pruning_min_kpts": -1}
    sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    case = dict(gc.LG_CASES["fixed"], m=2048, n=2048, seed=21)
    f0, f1 = gc.lg_inputs(case)
    ref32 = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = lightglue_ref.lightglue_forward(f0["kpts"].double(), f0["desc"].double(), f0["size"].double(), f1["kpts"].double(),
                                            f1["desc"].double(), f1["size"].double(), sd64, {**conf, "dtype": torch.float64}, taps=True)
    net = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
    data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
    dense = net(data, dense=True)["dense"].cpu().double()
    la64 = ref64["log_assignment"][:2048, :2048]
    e_ref = (ref32["log_assignment"][:2048, :2048].double() - la64).abs().max().item()
    e_hip = (dense[:2048, :2048] - la64).abs().max().item()
    print(f"log-assignment max error vs fp64: reference-equivalent fp32 CPU {e_ref:.3e}, HIP fp16x3 {e_hip:.3e}")
    assert e_hip <= 1e-3 and e_hip <= max(2.0 * e_ref, 2e-4)
