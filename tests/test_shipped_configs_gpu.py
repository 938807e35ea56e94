"""GPU (MI355X): the reference's SHIPPED configurations above 4096 keypoints.

  config/aliked.yaml:10-15            extractor aliked, model aliked-n16rot, max_num_keypoints 8000, detection_threshold 0.2, nms_radius 3,
                                      general.tile_size (2000, 2000); matcher lightglue 0.95 / 0.99 / 0.10
  config/superpoint+superglue.yaml:8-13   extractor superpoint, max_keypoints 8000, nms_radius 4, keypoint_threshold 0.005, remove_borders 4

through the plugin hooks (`AlikedExtractor._extract`, `SuperPointExtractor._extract`, `LightGlueMatcher._match_pairs`) on one 2000 x 2000 tile
of real photographs (tests/golden_cases.real_mosaic), against the oracle (pinned to the reference modules by oracle/make_golden.py).  The
values quoted here are the files' (tests/test_reference_yaml_configs.py reads the files themselves where /root/reference exists)."""
import importlib
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import aliked_ref, lightglue_ref, superpoint_ref
from tests import golden_cases as gc
from tests.parity import compare_superpoint, match_list_difference_is_a_tie
from tests.test_aliked_emu import compare_aliked

pytestmark = pytest.mark.gpu
ALIKED_CKPT = Path(__file__).parent / "assets" / "aliked-n16rot.pth"
ALIKED_YAML = {"model_name": "aliked-n16rot", "max_num_keypoints": 8000, "detection_threshold": 0.2, "nms_radius": 3}     # config/aliked.yaml:10-15
LG_YAML = {"n_layers": 9, "depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.10}                     # config/aliked.yaml:17-22
SP_YAML = {"max_keypoints": 8000, "nms_radius": 4, "keypoint_threshold": 0.005, "remove_borders": 4, "fix_sampling": False}  # config/superpoint+superglue.yaml:8-13
TILE = 2000                                                                                                                  # config/aliked.yaml:4


def _m(name):
    return importlib.import_module("deep-image-matching_amd." + name)


def _record(obj):
    d = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        d.mkdir(exist_ok=True)
        with open(d / "parity_measured.jsonl", "a") as f:
            f.write(json.dumps(obj) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def mosaic():
    return gc.real_mosaic(TILE + 48, TILE + 64)


def test_superpoint_8000_keypoints_on_a_2000_tile_vs_oracle(hip_lib, mosaic):
    """max_keypoints 8000 > the one-workgroup top-k's 4096: radix select -> chunk sorts -> rank merge (sp_post.hip).  Keypoint set identical
    (ties at the 8000th score excepted), score-descending order, scores 1e-5, descriptors 1e-3."""
    a = mosaic[:TILE, :TILE].astype(np.int64)
    gray = ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(np.uint8).astype(np.float32)   # Q5 (golden_cases.real_gray)
    ex = _m("plugins").SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", **SP_YAML, "allow_synthetic_weights": True}})
    f = ex._extract(gray)
    assert f["keypoints"].shape == (8000, 2) and f["descriptors"].shape == (256, 8000) and f["scores"].shape == (8000,)
    out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}
    ref = superpoint_ref.superpoint_forward(torch.tensor(gray / 255.0, dtype=torch.float)[None, None], ex._sd, SP_YAML)
    assert ref["keypoints"].shape[0] == 8000
    res = compare_superpoint(out, ref)
    assert res["common"] >= 7990, res
    assert bool((out["scores"][:-1] >= out["scores"][1:]).all())        # SPN:74-78: torch.topk's order
    # a second, different image through the same handle: the key table is rewritten, not accumulated
    g2 = np.ascontiguousarray(gray[::-1, ::-1])
    f2 = ex._extract(g2)
    ref2 = superpoint_ref.superpoint_forward(torch.tensor(g2 / 255.0, dtype=torch.float)[None, None], ex._sd, SP_YAML)
    res2 = compare_superpoint({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f2.items()}, ref2)
    _record({"test": "shipped_superpoint_8000", **{k: v for k, v in res.items()}, "second_image_common": res2["common"]})


@pytest.fixture(scope="module")
def aliked_views(hip_lib, mosaic):
    """Two overlapping 2000 x 2000 views of the mosaic (offset (40, 24)) through AlikedExtractor._extract with config/aliked.yaml's values and the
    trained aliked-n16rot checkpoint."""
    if not ALIKED_CKPT.exists():
        pytest.skip("aliked-n16rot.pth asset not present")
    ex = _m("plugins").AlikedExtractor({"general": {"tile_size": (TILE, TILE)}, "extractor": {"name": "aliked", **ALIKED_YAML, "weights_path": str(ALIKED_CKPT)}})
    views = [mosaic[:TILE, :TILE], mosaic[24:24 + TILE, 40:40 + TILE]]
    feats = [ex._extract(np.ascontiguousarray(v).astype(np.float32)) for v in views]
    return ex, views, feats


def test_aliked_yaml_8000_keypoints_trained_checkpoint_vs_oracle(hip_lib, aliked_views):
    ex, views, feats = aliked_views
    f = feats[0]
    assert f["keypoints"].shape == (8000, 2) and f["descriptors"].shape == (128, 8000)
    out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}
    img = torch.tensor(views[0].astype(np.float32).transpose(2, 0, 1)[None] / 255.0, dtype=torch.float)
    ref = aliked_ref.aliked_forward(img, ex._sd, ALIKED_YAML, taps=True)
    assert ref["keypoints"].shape[0] == 8000
    # sub-pixel keypoints: 5e-3 px here instead of the small images' 1e-3.  At 2000 x 2000 the REFERENCE's own fp32 evaluation sits 2.3e-3 px (score
    # map 7.6e-4) from an fp64 evaluation of the same module (train-mode BatchNorm statistics over 4 M pixels in fp32), the device 1.2e-4 px
    # (8.2e-5) — profiles/r06_aliked_tile_accuracy.json, scripts/gpu_aliked_tile_accuracy.py: the comparison measures the oracle's rounding
    res = compare_aliked(out, ref, ref_score_map=ref["score_map"], threshold=0.2, nms_radius=3, n_limit=8000, tie_tol=1e-3, kp_tol=5e-3)
    assert res["common"] >= 8000 - 8, {k: v for k, v in res.items() if k != "one_sided"}
    total, sites = _m("capi").saturation(hip_lib, None, reset=True)
    assert total == 0, ("fp16x3 range guard fired", sites)
    _record({"test": "shipped_aliked_yaml_8000", **{k: v for k, v in res.items() if k != "one_sided"}})


def test_aliked_yaml_features_through_lightglue_8000_x_8000_vs_oracle(hip_lib, aliked_views):
    """The float16 features features.h5 would hold (EB:60-67) of both views -> LightGlueMatcher._match_pairs with config/aliked.yaml's matcher
    values -> (S, 2) index pairs, vs the oracle on the same arrays: equal, or different only at numerical ties of the oracle's own assignment."""
    ex, views, feats = aliked_views
    weights = _m("weights")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 128)
    mt = _m("plugins").LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", **LG_YAML, "allow_synthetic_weights": True}}, local_features="aliked")
    mt._sd = sd
    fa, fb = ({**gc.fp16_round_trip(f), "image_size": np.array([TILE, TILE], np.int32)} for f in feats)
    m = torch.from_numpy(mt._match_pairs(fa, fb))
    assert m.dtype == torch.int64 and m.shape[1] == 2
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))  # noqa: E731
    conf = {k: LG_YAML[k] for k in ("depth_confidence", "width_confidence", "filter_threshold")}
    o = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                        t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, conf, taps=True)
    n_ref = int(o["matches"].shape[0])
    assert n_ref > 1000, n_ref
    flips = [] if torch.equal(m, o["matches"]) else match_list_difference_is_a_tie(m, o["matches"], o["log_assignment"], 0.1, tie_tol=3.6e-4,
                                                                                   ind0=o.get("ind0"), ind1=o.get("ind1"))
    assert len(flips) <= 3 * 9.6e-5 * n_ref + 1, flips
    total, sites = _m("capi").saturation(hip_lib, None, reset=True)
    assert total == 0, sites
    _record({"test": "shipped_aliked_yaml_lightglue_8000x8000", "reference_matches": n_ref, "device_matches": int(m.shape[0]), "explained_near_ties": len(flips),
             "stop": int(o["stop"])})
