"""GPU (MI355X): BASELINE configs[0] on its REAL inputs through the plugin hooks a user of the reference gets.

The five sacre-coeur photographs (and the reference's three pytest photographs) -> `SuperPointExtractor._extract` with
config/superpoint+lightglue.yaml's parameters -> `LightGlueMatcher._match_pairs` on the float16 features the reference would have read back
from features.h5, for the 10 brute-force pairs; `AlikedExtractor._extract` with the TRAINED aliked-n16rot checkpoint on the RGB photographs ->
LightGlue (128-d).  Everything is compared with the REFERENCE MODULES' OWN recorded outputs (tests/golden/config1_*.npz); the fp16x3 range
guard must stay silent on real photographs (policy "raise": a trip fails the test instead of silently re-running in bf16x6)."""
import importlib
import json
from itertools import combinations
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import aliked_ref, lightglue_ref
from tests import golden_cases as gc
from tests.config1_real import check_pixels, compare_sparse, gold, golden_features, lg_golden, stem
from tests.parity import compare_lightglue, match_list_difference_is_a_tie

pytestmark = pytest.mark.gpu
ALIKED_CKPT = Path(__file__).parent / "assets" / "aliked-n16rot.pth"


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
def sp_plugin(hip_lib):
    return _m("plugins").SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", **gc.CONFIG1_SP, "allow_synthetic_weights": True,
                                                                             "on_saturation": "raise"}})


def test_superpoint_hook_on_the_real_photographs_vs_the_reference_module(hip_lib, sp_plugin):
    capi = _m("capi")
    g = gold("sp")
    worst = {"score": 0.0, "desc": 0.0, "desc_proj": 0.0}
    for n in gc.SACRE_COEUR + gc.PYTEST_IMAGES:
        gray = gc.real_gray(n)
        check_pixels(g, n, gray)
        f = sp_plugin._extract(gray)                                         # float32 (H, W), 0..255: what ExtractorBase.extract passes (EB:197-207)
        assert f["keypoints"].dtype == np.float32 and f["descriptors"].shape == (256, 2000) and f["scores"].shape == (2000,)
        out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}
        res = compare_sparse(out, g, n, 256, subpixel=False, score_tol=1e-5)
        # keypoint SET: identical, except numerical ties at the top-k cut (the 2000th score), each within 1e-4 of it
        cut = float(min(out["scores"].min(), g[stem(n) + "/scores"].min()))
        for i in res["only_out"]:
            assert abs(float(out["scores"][i]) - cut) <= 1e-4, (n, i, res)
        for j in res["only_ref"]:
            assert abs(float(g[stem(n) + "/scores"][j]) - cut) <= 1e-4, (n, j, res)
        assert res["n_out"] == res["n_ref"] == 2000 and res["common"] >= 1996, (n, {k: v for k, v in res.items()})
        assert bool((out["scores"][:-1] >= out["scores"][1:]).all())        # top-k order: score-descending (SPN:74-78)
        for k in worst:
            worst[k] = max(worst[k], res[k])
    total, sites = capi.saturation(hip_lib, None, reset=True)
    assert total == 0, ("fp16x3 range guard fired on a real photograph", sites)
    _record({"test": "config1_real_superpoint_hook", "images": 8, **worst})


@pytest.mark.parametrize("variant", ["generic", "generic_t0", "matching"])
def test_lightglue_hook_on_the_reference_float16_features_all_10_pairs(hip_lib, variant):
    """_match_pairs(feats0, feats1) with the numpy float16 groups of features.h5 (MB:221-222, LGX:102-125) -> (S, 2) index pairs, and the
    full result dict of the resident matcher, vs the reference module's outputs on the same arrays."""
    weights = _m("weights")
    g = gold("lg")
    center = torch.as_tensor(g["center"])
    th = 0.0 if variant == "generic_t0" else 0.1
    mt = _m("plugins").LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "filter_threshold": th, "pruning_min_kpts": -1,
                                                                     "allow_synthetic_weights": True, "on_saturation": "raise"}}, local_features="superpoint")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 256, center=center) if variant == "matching" else weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    mt._sd = sd
    conf = dict(gc.CONFIG1_LG, filter_threshold=th)
    total_matches, worst, ties_seen = 0, 0.0, []
    for a, b in gc.config1_pairs():
        fa, fb = golden_features("superpoint", gc.SACRE_COEUR[a]), golden_features("superpoint", gc.SACRE_COEUR[b])
        ref = lg_golden(g, f"{variant}/{a}_{b}")
        m = mt._match_pairs(dict(fa, tile_idx=np.zeros(2000, np.float16)), dict(fb, tile_idx=np.zeros(2000, np.float16)))
        assert m.dtype == np.int64 and m.shape[1] == 2
        t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
        data = {"image0": {"keypoints": t(fa["keypoints"])[None], "descriptors": t(fa["descriptors"]).t()[None].contiguous(), "image_size": t(fa["image_size"])[None]},
                "image1": {"keypoints": t(fb["keypoints"])[None], "descriptors": t(fb["descriptors"]).t()[None].contiguous(), "image_size": t(fb["image_size"])[None]}}
        res = mt._net(data, dense=True)
        res = {k: ([x.cpu() for x in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in res.items()}
        assert np.array_equal(m, res["matches"][0].numpy())
        dense_ref = None
        if not (torch.equal(res["matches0"].reshape(-1).long(), ref["matches0"]) and torch.equal(res["matches1"].reshape(-1).long(), ref["matches1"])):
            # a differing decision must be a numerical tie of the REFERENCE's own assignment: the oracle (== reference on this pair, asserted by
            # the generator) supplies the dense log-assignment the reference module does not return
            o = lightglue_ref.lightglue_forward(data["image0"]["keypoints"][0], data["image0"]["descriptors"][0], data["image0"]["image_size"][0],
                                                data["image1"]["keypoints"][0], data["image1"]["descriptors"][0], data["image1"]["image_size"][0], sd, conf, taps=True)
            dense_ref = o["log_assignment"]
        info = compare_lightglue(res, ref, dense_ref=dense_ref, dense_out=res.get("dense"), filter_threshold=th, tie_tol=3e-4)
        ties_seen += info.get("explained_near_ties", [])
        worst = max(worst, info["max_matching_scores0_diff"], info["max_matching_scores1_diff"])
        total_matches += int(ref["matches"].shape[0])
    if variant != "generic":
        assert total_matches > 0
    capi = _m("capi")
    total, sites = capi.saturation(hip_lib, None, reset=True)
    assert total == 0, sites
    _record({"test": "config1_real_lightglue_hook", "variant": variant, "pairs": 10, "reference_matches": total_matches, "max_score_diff": worst,
             "explained_near_ties": len(ties_seen)})


def test_superpoint_hook_features_through_the_lightglue_hook_real_match_lists_on_the_dsc_photographs(hip_lib, sp_plugin):
    """the SuperPoint -> LightGlue leg of config 1 with REAL match lists.  The three overlapping DSC photographs; (a) the
    reference's float16 features through `_match_pairs` with matching-capable weights whitened on those descriptors: 132 / 180 / 140 reference
    matches at the default threshold, index pairs equal (or a numerical tie of the reference's own assignment); (b) the DEVICE's own `_extract`
    features of the same photographs (float16 round trip, as features.h5 stores them) through the same hook: the same keypoints match (compared
    by pixel coordinates, since the top-k order among equal scores is free)."""
    weights = _m("weights")
    g = gold("lg")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 256, sharpness=2.0, center=torch.as_tensor(g["dsc/center"]), whiten=torch.as_tensor(g["dsc/whiten"]))
    mt = _m("plugins").LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "pruning_min_kpts": -1, "allow_synthetic_weights": True,
                                                                     "on_saturation": "raise"}}, local_features="superpoint")
    mt._sd = sd
    own = {n: {**gc.fp16_round_trip(sp_plugin._extract(gc.real_gray(n))), "image_size": np.array(gc.real_gray(n).shape[:2], np.int32)} for n in gc.PYTEST_IMAGES}
    n_ref, flips, same_px = 0, [], []
    for na, nb in combinations(gc.PYTEST_IMAGES, 2):
        fa, fb = golden_features("superpoint", na), golden_features("superpoint", nb)
        ref = lg_golden(g, f"dsc/{stem(na)}__{stem(nb)}")
        m = torch.from_numpy(mt._match_pairs(fa, fb))
        n_ref += int(ref["matches"].shape[0])
        assert ref["matches"].shape[0] >= 100
        if not torch.equal(m, ref["matches"]):
            t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))  # noqa: E731
            o = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                                t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, dict(gc.CONFIG1_LG), taps=True)
            flips += match_list_difference_is_a_tie(m, ref["matches"], o["log_assignment"], 0.1, tie_tol=3.6e-4, ind0=o.get("ind0"), ind1=o.get("ind1"))
        # (b) end to end on the device: extract -> float16 -> match; as pixel-coordinate pairs
        mo = mt._match_pairs(own[na], own[nb])
        px = lambda f, idx: [tuple(int(v) for v in f["keypoints"][i]) for i in idx]  # noqa: E731
        got = set(zip(px(own[na], mo[:, 0]), px(own[nb], mo[:, 1])))
        want = set(zip(px(fa, ref["matches"][:, 0].tolist()), px(fb, ref["matches"][:, 1].tolist())))
        same_px.append((len(got & want), len(want)))
        assert len(got & want) >= 0.95 * len(want) and len(got) <= 1.05 * len(want), (na, nb, len(got), len(want), len(got & want))
    assert len(flips) <= 1, flips
    total, sites = _m("capi").saturation(hip_lib, None, reset=True)
    assert total == 0, sites
    _record({"test": "config1_real_superpoint_lightglue_dsc", "pairs": 3, "reference_matches": n_ref, "explained_near_ties": len(flips),
             "device_end_to_end_common_of_reference": same_px})


@pytest.fixture(scope="module")
def aliked_plugin(hip_lib):
    if not ALIKED_CKPT.exists():
        pytest.skip("aliked-n16rot.pth asset not present")
    return _m("plugins").AlikedExtractor({"general": {}, "extractor": {"name": "aliked", **gc.CONFIG1_AL, "weights_path": str(ALIKED_CKPT)}})


def test_aliked_hook_trained_checkpoint_on_the_real_photographs_vs_the_reference_module(hip_lib, aliked_plugin):
    """Real weights on real pixels: the checkpoint the reference ships, the photographs the reference ships, the reference module's outputs."""
    capi = _m("capi")
    g = gold("aliked")
    sd = aliked_plugin._sd
    worst = {"kp": 0.0, "score": 0.0, "desc": 0.0, "desc_proj": 0.0}
    near = 0
    capi.saturation(hip_lib, None, reset=True)
    for n in gc.SACRE_COEUR + gc.PYTEST_IMAGES:
        rgb = gc.real_rgb(n)
        check_pixels(g, n, rgb)
        f = aliked_plugin._extract(rgb.astype(np.float32))                   # float32 (H, W, 3) RGB 0..255 (EB:190-202, grayscale = False)
        out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}
        assert out["descriptors"].shape[0] == 128
        res = compare_sparse(out, g, n, 128, subpixel=True, score_tol=1e-3)
        if res["only_out"] or res["only_ref"]:
            # explained from the reference's own score map (the oracle is bit-exactly the reference on these inputs): threshold / NMS ties
            img = torch.tensor(rgb.astype(np.float32).transpose(2, 0, 1)[None] / 255.0, dtype=torch.float)
            o = aliked_ref.aliked_forward(img, sd, gc.CONFIG1_AL, taps=True)
            sm = o["score_map"].reshape(o["score_map"].shape[-2], o["score_map"].shape[-1])

            def explained(xy):
                x, y = int(round(float(xy[0]))), int(round(float(xy[1])))
                v = float(sm[max(0, y - 1): y + 2, max(0, x - 1): x + 2].max())
                win = sm[max(0, y - 3): y + 4, max(0, x - 3): x + 4].reshape(-1)
                top = torch.topk(win, 2).values
                return min(abs(v - 0.2), float(top[0] - top[1])) <= 2e-5
            for i in res["only_out"]:
                assert explained(out["keypoints"][i]), (n, "only_out", out["keypoints"][i].tolist())
            for j in res["only_ref"]:
                assert explained(g[stem(n) + "/keypoints"][j]), (n, "only_ref", g[stem(n) + "/keypoints"][j].tolist())
            near += max(len(res["only_out"]), len(res["only_ref"]))
        assert abs(res["n_out"] - res["n_ref"]) <= 2 and res["common"] >= res["n_ref"] - 2 and res["n_ref"] > 1000, (n, res["n_out"], res["n_ref"], res["common"])
        for k in worst:
            worst[k] = max(worst[k], res[k])
    total, sites = capi.saturation(hip_lib, None, reset=True)
    assert total == 0, ("fp16x3 range guard fired on a real photograph with the trained checkpoint", sites)
    assert near <= 6
    _record({"test": "config1_real_aliked_hook_trained", "images": 8, "near_tie_keypoints": near, **worst})


def test_lightglue_hook_on_the_trained_aliked_float16_features_real_matches(hip_lib):
    """13 pairs with 113 .. 1336 reference matches each at the default threshold 0.1."""
    weights = _m("weights")
    g = gold("aliked_lg")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 128)
    mt = _m("plugins").LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "pruning_min_kpts": -1, "allow_synthetic_weights": True,
                                                                     "on_saturation": "raise"}}, local_features="aliked")
    mt._sd = sd
    n_ref = n_same = 0
    flips = []
    for grp in (gc.PYTEST_IMAGES, gc.SACRE_COEUR):
        for na, nb in combinations(grp, 2):
            fa, fb = golden_features("aliked", na), golden_features("aliked", nb)
            ref = lg_golden(g, f"{stem(na)}__{stem(nb)}")
            m = torch.from_numpy(mt._match_pairs(fa, fb))
            n_ref += int(ref["matches"].shape[0])
            if torch.equal(m, ref["matches"]):
                n_same += int(m.shape[0])
                continue
            t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
            o = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                                t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, dict(gc.CONFIG1_LG), taps=True)
            flips += match_list_difference_is_a_tie(m, ref["matches"], o["log_assignment"], 0.1, tie_tol=3.6e-4, ind0=o.get("ind0"), ind1=o.get("ind1"))
    assert n_ref > 6000
    assert len(flips) <= 3 * 9.6e-5 * n_ref + 1, flips                     # the measured fp32-vs-fp64 flip rate of the reference itself
    capi = _m("capi")
    total, sites = capi.saturation(hip_lib, None, reset=True)
    assert total == 0, sites
    _record({"test": "config1_real_aliked_lightglue_hook", "pairs": 13, "reference_matches": n_ref, "explained_near_ties": len(flips)})


def test_batched_image_matcher_on_the_jpeg_files_equals_the_hooks(hip_lib, sp_plugin, tmp_path):
    """The five JPEG files -> BatchedImageMatcher.extract_features (default loader: PIL + the Q5 grey formula) -> features.h5 -> match_pairs ->
    raw_matches.h5: equal to the per-call hooks on the same files (the float16 round trip included)."""
    bm, export = _m("batched_matcher"), _m("export")
    mt = _m("plugins").LightGlueMatcher({"general": {"geom_verification": "NONE"}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "filter_threshold": 0.0,
                                                                                                "pruning_min_kpts": -1, "allow_synthetic_weights": True}},
                                        local_features="superpoint")
    paths = [gc.REAL_DIR / n for n in gc.SACRE_COEUR]
    shim = bm.BatchedImageMatcher(sp_plugin, mt, tmp_path, image_batch=4, pair_batch=4, verify=False)
    fpath = shim.extract_features(paths)
    pairs = [(paths[a].name, paths[b].name) for a, b in gc.config1_pairs()]
    shim.match_pairs(fpath, pairs)
    for n in gc.SACRE_COEUR:
        f = export.FeatureStore.read(fpath, n)
        h = sp_plugin._extract(gc.real_gray(n))
        # the ORDER among keypoints of equal score is not defined (torch.topk's is not either) and real photographs have
        # such ties (saturated regions): compare the keypoint SETS and each keypoint's descriptor
        hk, hd = h["keypoints"].astype(np.float16).astype(np.float32), h["descriptors"].astype(np.float16).astype(np.float32)
        ia = {tuple(k): i for i, k in enumerate(f["keypoints"].tolist())}
        ib = {tuple(k): i for i, k in enumerate(hk.tolist())}
        assert len(ia) == len(ib) == 2000 and set(ia) == set(ib), (n, len(set(ia) ^ set(ib)))
        perm = np.array([ib[tuple(k)] for k in f["keypoints"].tolist()])
        assert np.array_equal(f["descriptors"], hd[:, perm]), n
        same_order = bool(np.array_equal(f["keypoints"], hk))
        if not same_order:
            sc = h["scores"][perm]
            moved = np.nonzero((f["keypoints"] != hk).any(1))[0]
            assert all(np.isclose(h["scores"][i], sc[i], rtol=0, atol=0) for i in moved), (n, "order differs at keypoints of different score")
        assert tuple(int(v) for v in f["image_size"]) == gc.real_gray(n).shape[:2]
    raw = export.MatchStore.read_all(tmp_path / "raw_matches.h5")
    assert len(raw) == 10
    for a, b in pairs:
        hook = mt._match_pairs(export.FeatureStore.read(fpath, a), export.FeatureStore.read(fpath, b))
        assert np.array_equal(np.asarray(raw[(a, b)]).reshape(-1, 2), hook)
