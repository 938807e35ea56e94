"""GPU (MI355X): ALIKED HIP path through the C ABI vs the oracle / reference goldens, and ALIKED -> LightGlue."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import aliked_ref, lightglue_ref
from tests import golden_cases as gc
from tests.test_aliked_emu import compare_aliked

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _al():
    return importlib.import_module("deep-image-matching_amd.aliked_hip")


@pytest.mark.parametrize("name", list(gc.AL_CASES))
def test_aliked_gpu_vs_reference_golden(hip_lib, name):
    case = gc.AL_CASES[name]
    sd, img = gc.al_weights(case), gc.al_image(case)
    net = _al().AlikedHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=4096)
    out = {k: v.cpu() for k, v in net(img.cuda()).items()}
    g = np.load(GOLD / f"al_{name}.npz")
    gold = {k: torch.from_numpy(g[k]) for k in ("keypoints", "scores", "descriptors")}
    compare_aliked(out, gold, label=f"aliked golden {name} (reference module output)")


def test_aliked_gpu_tile_size_vs_oracle_and_batch(hip_lib):
    """A 384x512 RGB tile (the reference's tiling path feeds tiles like this), batch of 2 == singles."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_aliked_state_dict(7)
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 2000, "detection_threshold": 0.2, "nms_radius": 2}
    imgs = torch.rand(2, 3, 384, 512, generator=torch.Generator().manual_seed(2))
    net = _al().AlikedHIP(sd, cfg, max_batch=2, max_hw=(384, 512), capacity=2000)
    kp, sc, de, n = [t.cpu() for t in net.extract_batch(imgs.permute(0, 2, 3, 1).contiguous().cuda())]
    for b in range(2):
        ref = aliked_ref.aliked_forward(imgs[b][None], sd, cfg)
        k = int(n[b])
        out = {"keypoints": kp[b, :k], "scores": sc[b, :k], "descriptors": de[b, :k].t()}
        res = compare_aliked(out, ref, label=f"aliked 384x512 tile {b}, 2000 keypoints, HIP vs fp32 oracle")
        assert res["n_out"] == 2000
        single = {k_: v.cpu() for k_, v in net(imgs[b][None].cuda()).items()}
        assert torch.equal(single["keypoints"], out["keypoints"]) and torch.equal(single["descriptors"], out["descriptors"])


def test_aliked_plus_lightglue_gpu(hip_lib):
    """ALIKED (128-d) -> LightGlue with input_proj, vs the oracle chain on the SAME features."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    sd = weights.synthetic_aliked_state_dict(7)
    lsd = weights.synthetic_lightglue_state_dict(3, 128, gain=2.0)
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 300, "detection_threshold": 0.2, "nms_radius": 2}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    imgs = torch.rand(2, 200, 264, 3, generator=torch.Generator().manual_seed(4)).cuda()
    ext = _al().AlikedHIP(sd, cfg, max_batch=2, max_hw=(200, 264), capacity=300)
    mat = lg.LightGlueHIP(lsd, conf, max_pairs=1, max_kpts=300)
    kp, sc, de, n = ext.extract_batch(imgs)
    size = torch.tensor([[200.0, 264.0]] * 2, device="cuda")
    o = mat.match_batch(kp, de, n, size, n_pairs=1)
    S = int(o["n_matches"][0])
    k0, k1 = int(n[0]), int(n[1])
    ref = lightglue_ref.lightglue_forward(kp[0, :k0].cpu(), de[0, :k0].cpu(), size[0].cpu(), kp[1, :k1].cpu(), de[1, :k1].cpu(), size[1].cpu(), lsd, conf)
    assert int(o["stop"][0]) == ref["stop"]
    assert torch.equal(o["matches"][0, :S].cpu(), ref["matches"])
    assert (o["scores"][0, :S].cpu() - ref["scores"]).abs().max().item() < 1e-3 if S else True


# ---- the TRAINED checkpoint on hardware (VERDICT r3 missing #3 / next #3) ---------------------------------------------------
# aliked-n16rot.pth is the file the reference ships inside thirdparty/ALIKED/models (ALN:629-630 loads it from there); a byte
# copy travels with the tests as tests/assets/aliked-n16rot.pth (md5 bfec5e8086e9f6bf68ffeb90ca7a793a; ALIKED's BSD-3 licence),
# $DIM_ALIKED_WEIGHTS overrides the path.  Trained BatchNorm scales / score head = the dynamic range the seeded weights lack.
import os

ALIKED_CKPT = Path(os.environ.get("DIM_ALIKED_WEIGHTS", Path(__file__).parent / "assets" / "aliked-n16rot.pth"))


def _record(obj):
    import json
    d = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        d.mkdir(exist_ok=True)
        with open(d / "parity_measured.jsonl", "a") as f:
            f.write(json.dumps(obj) + "\n")
    except OSError:
        pass


def _structured_rgb(seed, H, W):
    """blobs + texture + noise, three slightly different channels: gives the trained detector thousands of maxima"""
    g = gc.sp_image({"seed": seed, "H": H, "W": W, "kind": "blobs"})[0, 0]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    tex = 0.15 * torch.sin(xx / 3.1) * torch.cos(yy / 4.3) + 0.1 * torch.sin((xx + yy) / 7.7)
    n = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed + 1000)) * 0.1
    return (torch.stack([g, g * 0.9 + 0.05, g * 0.8 + 0.1]) + tex[None] + n).clamp(0, 1)[None].contiguous()


@pytest.mark.skipif(not ALIKED_CKPT.exists(), reason="aliked-n16rot.pth asset not present")
def test_aliked_trained_checkpoint_full_tile_fp16x3_guard_silent(hip_lib):
    """Config 5's tile (1500 x 1000, DIM's zoo parameters: 4000 keypoints, threshold 0.2, radius 3) with the TRAINED weights, in the
    default fp16x3 arithmetic, vs the oracle at north_star's 1e-3 — and the range guard must stay SILENT (a trip would mean the
    production rate silently halves through the bf16x6 re-run)."""
    capi = importlib.import_module("deep-image-matching_amd.capi")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.load_aliked_state_dict(str(ALIKED_CKPT), model_name="aliked-n16rot")
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 3, "on_saturation": "raise"}
    img = _structured_rgb(31, 1000, 1500)
    net = _al().AlikedHIP(sd, cfg, max_batch=1, max_hw=(1000, 1500), capacity=4000)
    assert capi.get_arithmetic(hip_lib) == 2
    capi.saturation(hip_lib, net._stream(), reset=True)
    kp, sc, de, n = net.extract_batch(img[0].permute(1, 2, 0).contiguous().cuda()[None])     # unguarded call: read the counters ourselves
    total, sites = capi.saturation(hip_lib, net._stream(), reset=True)
    assert total == 0, ("fp16x3 range guard fired on the trained checkpoint", sites)
    k = int(n[0])
    out = {"keypoints": kp[0, :k].cpu(), "scores": sc[0, :k].cpu(), "descriptors": de[0, :k].t().cpu()}
    ref = aliked_ref.aliked_forward(img, sd, {k_: v for k_, v in cfg.items() if k_ != "on_saturation"}, taps=True)
    res = compare_aliked(out, ref, label="aliked TRAINED n16rot, 1500x1000 tile, 4000 keypoints, fp16x3 HIP vs fp32 oracle",
                         ref_score_map=ref["score_map"], n_limit=4000)
    assert res["n_out"] == 4000 and res.get("near_tie_keypoints", 0) <= 4, res
    _record({"test": "aliked_trained_full_tile", "guard_total": total, "guard_sites": sites, **{k_: v for k_, v in res.items() if k_ != "one_sided"}})


@pytest.mark.parametrize("model", ["aliked-n32", "aliked-t16"])
def test_aliked_variant_trained_checkpoint_vs_oracle_guard_silent(hip_lib, model):
    """The other geometries of ALN:573-579 (VERDICT r3 missing #5) with the checkpoints the reference ships (tests/assets: byte copies):
    aliked-n32 (32 deformable sample positions), aliked-t16 (8 / 16 / 32 / 64 channels, 64-d descriptors); 640 x 480, 2048
    keypoints, default arithmetic vs the oracle (pinned to the reference's aliked.py with these files by oracle/make_golden.py)."""
    ckpt = ALIKED_CKPT.parent / f"{model}.pth"
    if not ckpt.exists():
        pytest.skip(f"{model}.pth asset not present")
    capi = importlib.import_module("deep-image-matching_amd.capi")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.load_aliked_state_dict(str(ckpt), model_name=model)
    cfg = {"model_name": model, "max_num_keypoints": 2048, "detection_threshold": 0.2, "nms_radius": 2, "on_saturation": "raise"}
    img = _structured_rgb(33, 480, 640)
    net = _al().AlikedHIP(sd, cfg, max_batch=1, max_hw=(480, 640), capacity=2048)
    capi.saturation(hip_lib, net._stream(), reset=True)
    kp, sc, de, n = net.extract_batch(img[0].permute(1, 2, 0).contiguous().cuda()[None])
    total, sites = capi.saturation(hip_lib, net._stream(), reset=True)
    assert total == 0, (f"fp16x3 range guard fired on the trained {model} checkpoint", sites)
    k = int(n[0])
    out = {"keypoints": kp[0, :k].cpu(), "scores": sc[0, :k].cpu(), "descriptors": de[0, :k].t().cpu()}
    ref = aliked_ref.aliked_forward(img, sd, {k_: v for k_, v in cfg.items() if k_ != "on_saturation"}, taps=True)
    assert out["descriptors"].shape[0] == (64 if model == "aliked-t16" else 128)
    res = compare_aliked(out, ref, label=f"{model} TRAINED, 640x480, 2048 keypoints, HIP (default arithmetic) vs fp32 oracle",
                         ref_score_map=ref["score_map"], n_limit=2048, nms_radius=2)
    assert res["n_out"] > 500 and res.get("near_tie_keypoints", 0) <= 4, res
    _record({"test": f"{model}_trained", "guard_total": total, **{k_: v for k_, v in res.items() if k_ != "one_sided"}})


@pytest.mark.skipif(not ALIKED_CKPT.exists(), reason="aliked-n16rot.pth asset not present")
def test_aliked_trained_checkpoint_then_lightglue(hip_lib):
    """Trained ALIKED features of an image and a shifted copy (true correspondences) -> LightGlue (128-d input_proj; seeded weights:
    no LightGlue checkpoint exists offline) vs the oracle chain on the SAME features; the guard stays silent end to end."""
    capi = importlib.import_module("deep-image-matching_amd.capi")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    sd = weights.load_aliked_state_dict(str(ALIKED_CKPT), model_name="aliked-n16rot")
    lsd = weights.synthetic_lightglue_state_dict(3, 128, gain=2.0)
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 1024, "detection_threshold": 0.2, "nms_radius": 3}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    base = _structured_rgb(32, 480 + 16, 640 + 24)[0]
    imgs = torch.stack([base[:, :480, :640], base[:, 16:, 24:]]).permute(0, 2, 3, 1).contiguous().cuda()
    ext = _al().AlikedHIP(sd, cfg, max_batch=2, max_hw=(480, 640), capacity=1024)
    mat = lg.LightGlueHIP(lsd, conf, max_pairs=1, max_kpts=1024)
    capi.saturation(hip_lib, ext._stream(), reset=True)
    kp, sc, de, n = ext.extract_batch(imgs)
    size = torch.tensor([[480.0, 640.0]] * 2, device="cuda")
    o = mat.match_batch(kp, de, n, size, n_pairs=1)
    total, sites = capi.saturation(hip_lib, ext._stream(), reset=True)
    assert total == 0, sites
    S = int(o["n_matches"][0])
    k0, k1 = int(n[0]), int(n[1])
    assert k0 > 500 and k1 > 500
    ref = lightglue_ref.lightglue_forward(kp[0, :k0].cpu(), de[0, :k0].cpu(), size[0].cpu(), kp[1, :k1].cpu(), de[1, :k1].cpu(), size[1].cpu(), lsd, conf,
                                          taps=True)
    from tests.parity import match_list_difference_is_a_tie
    assert int(o["stop"][0]) == ref["stop"]
    ties = match_list_difference_is_a_tie(o["matches"][0, :S].cpu(), ref["matches"], ref["log_assignment"], 0.0, tie_tol=1e-4, ind0=ref["ind0"], ind1=ref["ind1"])
    assert len(ties) <= 2, ties
    _record({"test": "aliked_trained_then_lightglue", "kpts": [k0, k1], "matches": S, "ref_matches": int(ref["matches"].shape[0]), "ties": ties})
