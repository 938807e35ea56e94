"""GPU (MI355X): BASELINE config 5 end to end — pipeline.TiledPairPipeline on 6000 x 4000 RGB images, 4 x 4 tiles of 1500 x 1000, ALIKED +
LightGlue, tile PRESELECTION on the device — against the ORACLE chain: oracle/tile_ref's resize / votes / selection /
sequential tile-pair loop (the reference's _match_by_tile, MB:362-485) driven by the oracle SuperPoint and LightGlue, on the features the
pipeline extracted.  The extraction itself is pinned elsewhere (ALIKED tile vs oracle, device merge vs ExtractorBase._extract_by_tile)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref, tile_ref

pytestmark = pytest.mark.gpu
PRE = 750      # 6000 x 4000 -> 750 x 500: an exact 8 x 8 box down-sampling, so crops at multiples of 64 px down-sample to shifted copies


def _m(name):
    return importlib.import_module("deep-image-matching_amd." + name)


def test_tiled_pair_pipeline_at_6000x4000_vs_the_oracle_chain(hip_lib):
    plugins, pl, tm, weights = _m("plugins"), _m("pipeline"), _m("tile_matching"), _m("weights")
    general = {"tile_size": (1500, 1000), "tile_overlap": 0, "tile_preselection_size": PRE, "min_matches_per_tile": 5, "quality": "HIGH",
               "allow_synthetic_weights": True}
    kp_tile = 512        # keypoints per tile: keeps the oracle's LightGlue at 512 x 512 per tile pair (the 4000-keypoint size is bench.py --workload config5)
    ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": kp_tile,
                                                                     "detection_threshold": 0.2, "nms_radius": 3, "allow_synthetic_weights": True}})
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}
    mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", **conf, "pruning_min_kpts": -1, "allow_synthetic_weights": True}},
                                  local_features="aliked")
    lg_sd = weights.synthetic_lightglue_matching_state_dict(0, 128)
    mt._sd = lg_sd
    rng = np.random.default_rng(9)
    canvas = rng.integers(0, 256, (4000 + 256, 6000 + 256, 3), dtype=np.uint8)
    # first band (the one tile_selection reads) = noise at 8 x 8-block scale: the down-sampled image keeps its contrast (see bench.py run_config5)
    canvas[..., 0] = np.kron(rng.integers(0, 256, ((4000 + 256) // 8, (6000 + 256) // 8), dtype=np.uint8), np.ones((8, 8), np.uint8))
    offs = [(0, 0), (192, 128), (64, 256)]
    images = [np.ascontiguousarray(canvas[dy:dy + 4000, dx:dx + 6000]).astype(np.float32) for dy, dx in offs]
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    pre0 = tm.TilePreselector(sp_sd, weights.synthetic_lightglue_state_dict(0, 256), tile_preselection_size=PRE, device="cuda", lib=hip_lib)
    f0 = pre0.features("warm", np.ascontiguousarray(images[0][..., 0]), "HIGH")
    center = f0[1][0, : int(f0[2][0])].mean(0).cpu()
    pre_sd = weights.synthetic_lightglue_matching_state_dict(0, 256, center=center)
    mt._tile_preselector = tm.TilePreselector(sp_sd, pre_sd, tile_preselection_size=PRE, device="cuda", lib=hip_lib)
    pipe = pl.TiledPairPipeline(ex, mt, 0, 1, selection="PRESELECTION", empty_selection_fallback="GRID", tile_pair_batch=16)
    feats = pipe.extract_all(images)                                  # numpy dicts (and the device tables match_all uses for this very list)
    assert all(int(f["keypoints"].shape[0]) > 12 * kp_tile and set(np.unique(f["tile_idx"]).astype(int)) == set(range(16)) for f in feats)
    pairs = pl.exhaustive_pairs(3)
    names = ["img0", "img1", "img2"]
    matches = pipe.match_all(images, feats, pairs, names=names)
    assert pipe.n_fallback == 0, "PRESELECTION selected nothing for some image pair"
    # ---- the oracle chain on the same features ----
    torch.set_num_threads(16)
    og = tm.tile_grid((4000, 6000), (1500, 1000), 0)
    assert og == tile_ref_grid()
    small = []
    for im in images:
        band = np.ascontiguousarray(im[..., 0])
        _, scale, new = tile_ref.preselection_sizes(band.shape, PRE)
        assert new == (750, 500) and abs(scale - 0.125) < 1e-15
        s = tile_ref.resize_area(band, new) / np.float32(255.0)
        o = superpoint_ref.superpoint_forward(torch.from_numpy(s)[None, None], sp_sd, dict(tm.PRESELECTION_SP_CONF))
        small.append((o, scale))

    def oracle_lg(sd, cf):
        def run(fa, fb):
            t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
            r = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]), t(fb["keypoints"]),
                                                t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, cf)
            return r["matches"].numpy()
        return run

    total = diff = n_tp = 0
    for (a, b), got in zip(pairs.tolist(), matches):
        (oa, sa), (ob, sb) = small[a], small[b]
        ka, kb = oa["keypoints"], ob["keypoints"]
        ea, eb = 1 + ka.max(0).values - ka.min(0).values, 1 + kb.max(0).values - kb.min(0).values     # no image_size in the reference's call (MB:1077-1079)
        r = lightglue_ref.lightglue_forward(ka, oa["descriptors"].t().contiguous(), ea, kb, ob["descriptors"].t().contiguous(), eb, pre_sd,
                                            dict(tm.PRESELECTION_LG_CONF))
        mm = r["matches"].numpy()
        assert mm.shape[0] > 100, (a, b, mm.shape)                  # the preselector really matches
        pa, pb = ka.numpy()[mm[:, 0]] / np.float32(sa), kb.numpy()[mm[:, 1]] / np.float32(sb)
        votes = tile_ref.tile_pair_votes(pa, pb, og, og, (1500, 1000))
        sel = tile_ref.select_tile_pairs("PRESELECTION", list(og), list(og), votes, 5)
        assert len(sel) > 16                                         # shifted crops: a tile overlaps up to four tiles of the other image
        assert sel == mt.tile_selection(names[a], names[b], "PRESELECTION", image0=np.ascontiguousarray(images[a][..., 0]),
                                        image1=np.ascontiguousarray(images[b][..., 0])), (a, b)
        want = tile_ref.match_by_tile(feats[a], feats[b], sel, oracle_lg(lg_sd, conf))
        n_tp += len(sel)
        total += int(want.shape[0])
        g, w = {tuple(x) for x in got.tolist()}, {tuple(x) for x in want.tolist()}
        diff += len(g ^ w)
        assert got.dtype == np.int64 and (got.shape[0] == 0 or np.array_equal(got, np.unique(got, axis=0)))     # sorted, unique: np.unique(axis=0) order
    assert total > 1000, total
    # the measured irreproducibility of the reference itself (fp32 vs fp64): 9.6e-5 of the matches, x 3, + 1
    import json
    from pathlib import Path
    try:
        with open(Path(__file__).resolve().parents[1] / "gpurun_out" / "parity_measured.jsonl", "a") as f:
            f.write(json.dumps({"test": "tiled_pipeline_6000x4000_vs_oracle_chain", "image_pairs": 3, "tile_pairs": n_tp, "oracle_matches": total,
                                "differing_matches": diff}) + "\n")
    except OSError:
        pass
    assert diff <= 3 * 9.6e-5 * total + 1, (diff, total)
    assert n_tp == pipe.timings["tile_pairs_total"]


def tile_ref_grid():
    """tile ids -> (x, y) origins of a 6000 x 4000 image cut into 1500 x 1000 windows without overlap (utils/tiling.py:62-192): no padding"""
    return {r * 4 + c: (c * 1500, r * 1000) for r in range(4) for c in range(4)}
