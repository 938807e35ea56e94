"""One command for whoever HAS the official checkpoints (they are URL downloads — superpoint.py:110 `superpoint_v1.pth`, lightglue.py:328
`superpoint_lightglue.pth` — and do not exist in the build container or on the GPU box):

    DIM_SP_WEIGHTS=/path/superpoint_v1.pth DIM_LG_WEIGHTS=/path/superpoint_lightglue.pth python scripts/check_official_weights.py [--oracle]

runs BASELINE configs[0] (the five sacre-coeur photographs of tests/assets/config1 -> 10 brute-force pairs, config/superpoint+lightglue.yaml:
nms 4 / threshold 0.005 / 2000 keypoints, LightGlue 0.95 / 0.99 / 0.10) and the three DSC photographs of the reference's pytest fixture through
the plugin hooks on an MI355X with the TRAINED weights and reports, per image / pair and in total:

  * keypoints, matches, stop layer;
  * the fp16x3 range guard (`dim_saturation_read`): which sites, if any, left the exact range of the fp16 split on trained weights — the
    headline bench has only ever been observed silent on seeded synthetic weights (VERDICT r5 missing #4); a hit means that call re-runs in
    bf16x6 (~1.4 x slower), nothing else;
  * with --oracle: the same images / pairs through oracle/*.py on the CPU (= the reference modules' arithmetic, pinned by
    oracle/make_golden.py) and the comparison the parity tests make (keypoint sets, descriptors 1e-3, match lists up to numerical ties).

Prints one JSON line (and writes gpurun_out/official_weights_report.json).  Exit code 1 when a comparison fails."""
import argparse
import importlib
import json
import os
import sys
from itertools import combinations
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle on every image / pair and compare")
    a = ap.parse_args()
    sp_path, lg_path = os.environ.get("DIM_SP_WEIGHTS"), os.environ.get("DIM_LG_WEIGHTS")
    if not sp_path or not lg_path:
        print("set DIM_SP_WEIGHTS=<superpoint_v1.pth> and DIM_LG_WEIGHTS=<superpoint_lightglue.pth> (the reference downloads them: superpoint.py:110, "
              "lightglue.py:328)", file=sys.stderr)
        return 2
    from tests import golden_cases as gc
    from tests.parity import compare_superpoint, match_list_difference_is_a_tie
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    capi = importlib.import_module("deep-image-matching_amd.capi")
    ex = plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", **gc.CONFIG1_SP, "weights_path": sp_path, "on_saturation": "fallback"}})
    mt = plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "weights_path": lg_path, "on_saturation": "fallback"}},
                                  local_features="superpoint")
    lib = ex._lib
    rep = {"images": {}, "pairs": {}, "range_guard": {}, "ok": True}
    feats = {}
    capi.saturation(lib, None, reset=True)
    for n in gc.SACRE_COEUR + gc.PYTEST_IMAGES:
        gray = gc.real_gray(n)
        # raw call first (no fallback) to SEE the guard, then the guarded hook for the result
        ex._ensure(*gray.shape)
        img = torch.from_numpy(gray / 255.0).to(ex._net.device, torch.float32)[None].contiguous()
        ex._net.extract_batch(img)
        total, sites = capi.saturation(lib, None, reset=True)
        f = ex._extract(gray)
        rep["images"][n] = {"keypoints": int(f["keypoints"].shape[0]), "range_guard_sites": sites}
        if total:
            rep["range_guard"][n] = sites
        feats[n] = {**gc.fp16_round_trip(f), "image_size": np.array(gray.shape[:2], np.int32)}
        if a.oracle:
            from oracle import superpoint_ref
            ref = superpoint_ref.superpoint_forward(torch.tensor(gray[None][None] / 255.0, dtype=torch.float), ex._sd, gc.CONFIG1_SP)
            try:
                res = compare_superpoint({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}, ref)
                rep["images"][n]["vs_oracle"] = res
            except AssertionError as e:
                rep["images"][n]["vs_oracle"] = {"FAILED": repr(e)[:300]}
                rep["ok"] = False
    for grp in (gc.SACRE_COEUR, gc.PYTEST_IMAGES):
        for na, nb in combinations(grp, 2):
            fa, fb = feats[na], feats[nb]
            mt._ensure(max(fa["keypoints"].shape[0], fb["keypoints"].shape[0]))
            pol, mt._net.on_saturation = mt._net.on_saturation, "off"
            mt._match_pairs(fa, fb)
            total, sites = capi.saturation(lib, None, reset=True)
            mt._net.on_saturation = pol
            m = mt._match_pairs(fa, fb)
            rec = {"matches": int(m.shape[0]), "range_guard_sites": sites}
            if total:
                rep["range_guard"][f"{na}|{nb}"] = sites
            if a.oracle:
                from oracle import lightglue_ref
                t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))  # noqa: E731
                o = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                                    t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), mt._sd, dict(gc.CONFIG1_LG), taps=True)
                rec["oracle_matches"], rec["stop"] = int(o["matches"].shape[0]), int(o["stop"])
                try:
                    ties = [] if torch.equal(torch.from_numpy(m), o["matches"]) else match_list_difference_is_a_tie(
                        torch.from_numpy(m), o["matches"], o["log_assignment"], 0.1, tie_tol=3.6e-4, ind0=o.get("ind0"), ind1=o.get("ind1"))
                    rec["explained_near_ties"] = len(ties)
                except AssertionError as e:
                    rec["FAILED"] = repr(e)[:300]
                    rep["ok"] = False
            rep["pairs"][f"{na}|{nb}"] = rec
    rep["range_guard_silent"] = not rep["range_guard"]
    rep["total_matches"] = sum(r["matches"] for r in rep["pairs"].values())
    print(json.dumps(rep))
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "official_weights_report.json").write_text(json.dumps(rep, indent=1))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
