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
