"""Shared helpers of the config-1 tests on the REAL photographs (tests/assets/config1, goldens tests/golden/config1_*.npz = the reference
modules' own outputs, recorded by oracle/make_golden.py main_config1)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from tests import golden_cases as gc

GOLD = Path(__file__).parent / "golden"
_cache = {}


def gold(name: str):
    if name not in _cache:
        _cache[name] = np.load(GOLD / f"config1_{name}.npz")
    return _cache[name]


def stem(n: str) -> str:
    return n.rsplit(".", 1)[0]


def check_pixels(g, n: str, arr):
    """The golden was made from exactly these pixels (PIL's decode of the committed JPEG bytes); another decoder -> not comparable."""
    want = str(g[stem(n) + "/pixels_sha1"])
    assert gc.pixel_digest(arr) == want, (f"{n}: this PIL decodes the JPEG to different pixels than the golden generator's did — regenerate "
                                          "tests/golden/config1_*.npz with oracle/make_golden.py config1")


def golden_features(kind: str, n: str) -> dict:
    """The float16 feature group save_features_h5 would have written for image `n` (kind 'superpoint' | 'aliked'), as numpy float16 arrays —
    what MatcherBase.match reads back (io/h5.py) and hands to _match_pairs."""
    g = gold("features_f16")
    p = f"{kind}/{stem(n)}/"
    return {"keypoints": g[p + "keypoints"], "descriptors": g[p + "descriptors"], "scores": g[p + "scores"],
            "image_size": g[p + "image_size"].astype(np.int32)}


def lg_golden(g, tag: str) -> dict:
    return {k: torch.as_tensor(g[f"{tag}/{k}"].astype(np.int64) if g[f"{tag}/{k}"].dtype.kind in "iu" else g[f"{tag}/{k}"])
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "matches", "scores", "prune0", "prune1")} | {"stop": int(g[f"{tag}/stop"])}


def compare_sparse(out: dict, g, n: str, dim: int, subpixel: bool, score_tol: float, desc_tol: float = 1e-3, kp_tol: float = 1e-3):
    """out: keypoints (N,2), scores (N,), descriptors (dim,N) CPU tensors of the implementation under test; g: config1_sp / config1_aliked
    golden (fp32 keypoints and scores of ALL reference keypoints, the full descriptor of every 16th, 4 fixed projections of every one).
    Keypoints are paired by integer pixel (SuperPoint) or nearest neighbour within 0.05 px (ALIKED's sub-pixel keypoints).  Returns the
    measured maxima and the unpaired indices of both sides (the caller explains or rejects those)."""
    s = stem(n)
    rk, rs = g[s + "/keypoints"].astype(np.float64), g[s + "/scores"]
    ok_, os_ = out["keypoints"].numpy().astype(np.float64), out["scores"].numpy()
    if subpixel:
        from scipy.spatial import cKDTree
        dist, nn = cKDTree(rk).query(ok_)
        pairs, used = [], set()
        for i, (d, j) in enumerate(zip(dist, nn)):
            if d <= 0.05 and int(j) not in used:
                used.add(int(j)); pairs.append((i, int(j)))
    else:
        ra = {(int(x), int(y)): j for j, (x, y) in enumerate(rk.tolist())}
        pairs = [(i, ra[(int(x), int(y))]) for i, (x, y) in enumerate(ok_.tolist()) if (int(x), int(y)) in ra]
    ia, ib = np.array([p[0] for p in pairs], dtype=np.int64), np.array([p[1] for p in pairs], dtype=np.int64)
    P = gc.desc_projection(dim)
    de = out["descriptors"].numpy()                                   # (dim, N)
    proj = de.T.astype(np.float64) @ P
    res = {"n_out": len(ok_), "n_ref": len(rk), "common": len(pairs),
           "kp": float(np.abs(ok_[ia] - rk[ib]).max()) if len(pairs) else 0.0,
           "score": float(np.abs(os_[ia] - rs[ib]).max()) if len(pairs) else 0.0,
           "desc_proj": float(np.abs(proj[ia] - g[s + "/desc_proj"][ib]).max()) if len(pairs) else 0.0}
    sub = [(i, j // gc.DESC_STRIDE) for i, j in pairs if j % gc.DESC_STRIDE == 0]
    ds = g[s + "/desc_sub"]                                           # (dim, ceil(N / 16))
    res["desc_sub_checked"] = len(sub)
    res["desc"] = float(np.abs(de[:, [i for i, _ in sub]] - ds[:, [j for _, j in sub]]).max()) if sub else 0.0
    res["only_out"] = sorted(set(range(len(ok_))) - set(ia.tolist()))
    res["only_ref"] = sorted(set(range(len(rk))) - set(ib.tolist()))
    assert res["kp"] <= kp_tol and res["score"] <= score_tol and res["desc"] <= desc_tol and res["desc_proj"] <= desc_tol, res
    return res
