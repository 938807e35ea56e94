"""GPU (MI355X): the BASELINE.json configurations that round 1 never parity-tested at their stated sizes
(VERDICT r1 next #1 a-c), each against the oracle, asserted.

config 1  config/superpoint+lightglue.yaml parameters (nms 4 / thr 0.005 / 2000 keypoints; LightGlue 0.95 / 0.99 / 0.1) on
          the five sacre-coeur image sizes (640x480, 618x640 x2, 640x618, 784x784: sides that are not multiples of 8, Q12),
          structured blobs+noise images, all 10 brute-force pairs with DIM's (H, W) image_size (Q4);
config 4  exhaustive pairs through PairMatchingPipeline: 24 images -> all 276 pairs, every pair's matches / stop / prune
          equal to the oracle on the same features;
config 5  ALIKED at a full 1500 x 1000 tile (4000 keypoints) vs the oracle, and the batched tile-pair matching vs the
          reference's sequential loop (oracle/tile_ref.match_by_tile) driven by the ORACLE LightGlue.
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import aliked_ref, lightglue_ref, superpoint_ref, tile_ref
from tests import golden_cases as gc
from tests.parity import compare_lightglue, compare_superpoint, match_list_difference_is_a_tie, order_is_reference_like
from tests.test_aliked_emu import compare_aliked

pytestmark = pytest.mark.gpu


def _m(name):
    return importlib.import_module("deep-image-matching_amd." + name)


def _cpu(res):
    return {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in res.items()}


SACRE_COEUR_HW = [(480, 640), (640, 618), (640, 618), (618, 640), (784, 784)]   # SURVEY §8(d) config 1 [probe]
YAML_SP = {"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": 2000, "remove_borders": 4, "fix_sampling": False}
YAML_LG = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": -1}


def test_config1_yaml_parameters_at_the_sacre_coeur_sizes(hip_lib):
    weights = _m("weights")
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    lg_sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    ext = _m("superpoint_hip").SuperPointHIP(sp_sd, YAML_SP, max_batch=1, max_hw=(784, 784), capacity=2000)
    feats = []
    for i, (H, W) in enumerate(SACRE_COEUR_HW):
        img = gc.sp_image({"seed": 40 + i, "H": H, "W": W, "kind": "blobs"})
        out = {k: v.cpu() for k, v in ext(img.cuda()).items()}
        ref = superpoint_ref.superpoint_forward(img, sp_sd, YAML_SP, taps=True)
        taps = ext.debug_taps()
        h8, w8 = (H // 8) * 8, (W // 8) * 8
        assert taps["score_map"].shape[-2:] == (h8, w8)
        assert (taps["score_map"][0] - ref["score_map"][0]).abs().max().item() < 1e-5
        assert torch.equal(superpoint_ref.simple_nms(taps["score_map"], 4)[0], taps["nms_map"][0])
        res = compare_superpoint(out, ref)
        assert res["n_out"] > 100
        order_is_reference_like(out, k_limited=(res["n_out"] == 2000))
        feats.append(out)
    # brute-force pairs at the shipped threshold (0.1) and at 0 (so that the match lists are not empty).  Round 5: four of the ten pairs (every
    # image and every size combination once) — all ten pairs run on the REAL photographs against the reference module's own outputs in
    # tests/test_config1_real_gpu.py, and the oracle's CPU time here was 118 s of the suite's 661
    n_matches = 0
    for th in (0.1, 0.0):
        conf = dict(YAML_LG, filter_threshold=th)
        mat = _m("lightglue_hip").LightGlueHIP(lg_sd, conf, max_pairs=1, max_kpts=2000)
        for a, b in ((0, 1), (1, 2), (2, 3), (3, 4)):
            for _ in (0,):
                sa, sb = torch.tensor(SACRE_COEUR_HW[a], dtype=torch.float32), torch.tensor(SACRE_COEUR_HW[b], dtype=torch.float32)  # (H, W): Q4
                ka, kb = feats[a]["keypoints"], feats[b]["keypoints"]
                da, db = feats[a]["descriptors"].t().contiguous(), feats[b]["descriptors"].t().contiguous()
                res = _cpu(mat({"image0": {"keypoints": ka[None], "descriptors": da[None], "image_size": sa[None]},
                                "image1": {"keypoints": kb[None], "descriptors": db[None], "image_size": sb[None]}}))
                ref = lightglue_ref.lightglue_forward(ka, da, sa, kb, db, sb, lg_sd, conf)
                compare_lightglue(res, ref)
                n_matches += ref["matches"].shape[0]
    assert n_matches > 0


def _record(obj):
    """measured numbers behind an assertion -> gpurun_out/parity_measured.jsonl (copied to profiles/ per round)"""
    import json
    from pathlib import Path
    d = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        d.mkdir(exist_ok=True)
        with open(d / "parity_measured.jsonl", "a") as f:
            f.write(json.dumps(obj) + "\n")
    except OSError:
        pass


def test_config4_exhaustive_pairs_through_the_pipeline_vs_oracle(hip_lib):
    weights, pl = _m("weights"), _m("pipeline")
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 400, "remove_borders": 4}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0, "pruning_min_kpts": -1}
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(1234), weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    n_img, H, W = 24, 200, 264
    imgs = torch.rand(n_img, H, W, generator=torch.Generator().manual_seed(4))
    ext = _m("superpoint_hip").SuperPointHIP(sp_sd, cfg, max_batch=8, max_hw=(H, W), capacity=400)
    mat = _m("lightglue_hip").LightGlueHIP(lg_sd, conf, max_pairs=16, max_kpts=400)
    pipe = pl.PairMatchingPipeline(ext, mat)
    table = pipe.extract_all(imgs.cuda())
    pairs = pl.exhaustive_pairs(n_img)
    assert pairs.shape == (276, 2)
    cnt, mt, ms, stop, prune = [t.cpu() for t in pipe.match_all(table, pairs, aux=True)]
    kp, _, de, n, size = [t.cpu() for t in table]
    total, ties = 0, []
    for p, (a, b) in enumerate(pairs.tolist()):
        na, nb = int(n[a]), int(n[b])
        ref = lightglue_ref.lightglue_forward(kp[a, :na], de[a, :na], size[a], kp[b, :nb], de[b, :nb], size[b], lg_sd, conf, taps=True)
        S = int(cnt[p])
        assert int(stop[p]) == ref["stop"], (p, a, b)
        assert torch.equal(prune[p, 0, :na].long(), ref["prune0"].long()) and torch.equal(prune[p, 1, :nb].long(), ref["prune1"].long()), (p, a, b)
        if torch.equal(mt[p, :S], ref["matches"]):
            if S:
                assert (ms[p, :S] - ref["scores"]).abs().max().item() <= 1e-3
        else:
            # 276 pairs x ~150 matches on random weights: a decision whose margin in the ORACLE's own log-assignment is below 1e-4
            # (fp32 LightGlue is not reproducible to that) may fall either way; anything else fails here
            # tolerance 3e-4 = the fp32 noise of ONE log-assignment entry (yardstick test: the reference-equivalent fp32 path is 2.7e-4
            # from an fp64 evaluation), i.e. about half of what two competing entries can move against each other; the oracle's own
            # margin at the one tie seen so far was 8.3e-5 on one host CPU and 8.6e-6 on another
            ties += match_list_difference_is_a_tie(mt[p, :S], ref["matches"], ref["log_assignment"], conf["filter_threshold"], tie_tol=3e-4, ind0=ref["ind0"], ind1=ref["ind1"])
            print("near-tie", (p, a, b), ties[-2:])
        total += S
    assert total > 276
    assert len(ties) <= 4, ties          # (measured in round 3: one tie, margin 8.3e-5, pair (17, 20))
    _record({"test": "config4_276_pairs", "matches_total": total, "explained_near_ties": ties})


def test_config5_aliked_full_tile_vs_oracle(hip_lib):
    """One 1500 x 1000 RGB tile (config 5's tile size; 1000 is not a multiple of 32: replicate padding), n_limit 4000."""
    weights = _m("weights")
    sd = weights.synthetic_aliked_state_dict(7)
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 3}
    img = torch.rand(1, 3, 1000, 1500, generator=torch.Generator().manual_seed(12))
    net = _m("aliked_hip").AlikedHIP(sd, cfg, max_batch=1, max_hw=(1000, 1500), capacity=4000)
    out = {k: v.cpu() for k, v in net(img.cuda()).items()}
    ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
    res = compare_aliked(out, ref, label="aliked 1500x1000 tile, 4000 keypoints, HIP vs fp32 oracle", ref_score_map=ref["score_map"], n_limit=4000)
    assert res["n_out"] == 4000 and res.get("near_tie_keypoints", 0) <= 2


def test_config5_batched_tile_matching_vs_the_sequential_loop_with_the_oracle_matcher(hip_lib):
    """3000 x 2000 RGB pair = 2 x 2 tiles of 1500 x 1000: batched ALIKED tile extraction through the plugin, then
    match_tile_pairs_batched (GRID + 2 cross pairs) vs oracle/tile_ref.match_by_tile whose per-tile-pair matcher is the
    ORACLE LightGlue (features="aliked": 128-d input_proj) on the same features."""
    plugins, tm, weights = _m("plugins"), _m("tile_matching"), _m("weights")
    general = {"tile_size": (1500, 1000), "tile_overlap": 0}
    ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 2000,
                                                                    "detection_threshold": 0.2, "nms_radius": 3, "allow_synthetic_weights": True}})
    mcfg = {"name": "lightglue", "depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0,
            "allow_synthetic_weights": True, "pruning_min_kpts": -1}
    mt = plugins.LightGlueMatcher({"general": general, "matcher": mcfg}, local_features="aliked")
    rng = np.random.default_rng(5)
    base = (rng.random((2000, 3000, 3)) * 255).astype(np.float32)
    feats = []
    for im in (base, np.roll(base, (37, 61), axis=(0, 1)).copy()):
        f = ex._extract_by_tile(im)
        f["image_size"] = np.array(im.shape[:2], np.int32)
        assert f["keypoints"].shape[0] > 4000 and set(np.unique(f["tile_idx"]).astype(int)) == {0, 1, 2, 3}
        feats.append(f)
    pairs = tm.select_tile_pairs("GRID", range(4), range(4)) + [(0, 1), (3, 2)]
    got = tm.match_tile_pairs_batched(mt._ensure_pairs, feats[0], feats[1], pairs, "cuda", pair_batch=4)

    def oracle_matcher(a, b):
        sz = torch.tensor([2000.0, 3000.0])
        r = lightglue_ref.lightglue_forward(torch.from_numpy(a["keypoints"]), torch.from_numpy(a["descriptors"].T.copy()), sz,
                                            torch.from_numpy(b["keypoints"]), torch.from_numpy(b["descriptors"].T.copy()), sz, mt._sd, {**mt._conf})
        return r["matches"].numpy()

    ref = tile_ref.match_by_tile(feats[0], feats[1], pairs, oracle_matcher)
    assert got.dtype == np.int64 and len(got) > 0 and np.array_equal(got, ref)


def flip_rate_basis():
    """The MEASURED basis of the near-tie rule: scripts/study/lg_flip_rate.py evaluated 200 pairs at
    2048 x 2048 keypoints with the oracle in fp32 (= the reference's arithmetic) and in fp64, and with the HIP path;
    profiles/r04_flip_rate_summary.json holds, for threshold 0 and 0.1, how often the REFERENCE ARITHMETIC ITSELF reports a
    different match than its own fp64 evaluation, and the largest fp64 decision margin at which it did.  A HIP-vs-oracle
    difference is accepted only inside that envelope: margin <= tie_tol = max(measured margin, 1e-4), count <= 3 x the
    measured rate x matches + 1."""
    import json
    from pathlib import Path
    prof = Path(__file__).resolve().parents[1] / "profiles"
    # round 4: 200 pairs, generic weights, fixed work: 41 535 matches at threshold 0 (4 flips fp32 vs fp64), 616 at 0.1;
    # round 5: 200 pairs of graded difficulty, matching-capable weights, adaptive depth / width ON: 343 588 matches at
    # threshold 0 and 305 948 at the reference's default 0.1 — 0 flips in every comparison (HIP vs fp64, HIP vs fp32, fp32 vs fp64), equal stop
    # layers on all 200 pairs.  The envelope is the WORST of the two studies.
    rate, tol, seen = 0.0, 1e-4, 0
    for name in ("r04_flip_rate_summary.json", "r05_flip_rate_summary.json"):
        if (prof / name).exists():
            s = json.loads((prof / name).read_text())
            ks = [k for k in s if k.startswith("threshold_")]
            rate = max([rate] + [s[k]["flip_rate_o32_vs_o64"] for k in ks])
            tol = max([tol] + [s[k]["max_margin_o32_vs_o64"] for k in ks])
            seen += 1
    return {"rate": rate, "tie_tol": tol, "studies": seen}


def test_config4_style_pairs_with_true_correspondences_vs_oracle(hip_lib):
    """config 4 with match lists that carry weight (the 276-pair test above has ~2.6 matches per pair): 10 crops
    of one canvas (multiple-of-8 shifts: SuperPoint features of a scene point are equal in every crop) -> all 45 pairs through
    PairMatchingPipeline with the matching-capable synthetic LightGlue weights at the reference's default threshold 0.1; EVERY pair
    must carry >= 100 matches, >= 90 % of them the true correspondences, and equal the oracle's list on the same features up to the
    measured near-tie envelope."""
    weights, pl, wl = _m("weights"), _m("pipeline"), _m("workloads")
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 512, "remove_borders": 4}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": -1}
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    n_img, H, W = 10, 320, 320
    imgs, off = wl.shifted_crops(n_img, H, W, max_shift=48, seed=3)
    ext = _m("superpoint_hip").SuperPointHIP(sp_sd, cfg, max_batch=10, max_hw=(H, W), capacity=512)
    lg_sd = weights.synthetic_lightglue_matching_state_dict(0, 256, center=wl.descriptor_mean(ext, imgs.cuda()))
    mat = _m("lightglue_hip").LightGlueHIP(lg_sd, conf, max_pairs=15, max_kpts=512)
    pipe = pl.PairMatchingPipeline(ext, mat)
    table = pipe.extract_all(imgs.cuda())
    pairs = pl.exhaustive_pairs(n_img)
    cnt, mt, ms, stop, prune = [t.cpu() for t in pipe.match_all(table, pairs, aux=True)]
    kp, _, de, n, size = [t.cpu() for t in table]
    basis = flip_rate_basis()
    total, ties, per_pair = 0, [], []
    for p, (a, b) in enumerate(pairs.tolist()):
        na, nb = int(n[a]), int(n[b])
        ref = lightglue_ref.lightglue_forward(kp[a, :na], de[a, :na], size[a], kp[b, :nb], de[b, :nb], size[b], lg_sd, conf, taps=True)
        S = int(cnt[p])
        assert S >= 100 and ref["matches"].shape[0] >= 100, (p, a, b, S)
        assert int(stop[p]) == ref["stop"], (p, a, b)
        assert torch.equal(prune[p, 0, :na].long(), ref["prune0"].long()) and torch.equal(prune[p, 1, :nb].long(), ref["prune1"].long()), (p, a, b)
        assert wl.true_match_fraction(kp[a], kp[b], mt[p, :S], off[a], off[b]) >= 0.9
        if torch.equal(mt[p, :S], ref["matches"]):
            assert (ms[p, :S] - ref["scores"]).abs().max().item() <= 1e-3
        else:
            ties += match_list_difference_is_a_tie(mt[p, :S], ref["matches"], ref["log_assignment"], conf["filter_threshold"], tie_tol=basis["tie_tol"],
                                                   ind0=ref["ind0"], ind1=ref["ind1"])
        total += S
        per_pair.append(S)
    assert len(ties) <= 3 * basis["rate"] * total + 1, (ties, basis)
    _record({"test": "config4_style_true_correspondences_45_pairs", "matches_total": total, "matches_per_pair_min": min(per_pair),
             "matches_per_pair_mean": total / len(per_pair), "explained_near_ties": ties, "basis": basis})
