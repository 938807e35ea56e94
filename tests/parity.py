"""Parity comparison helpers shared by the CPU (emulator) and GPU tests."""
from __future__ import annotations

import numpy as np
import torch


def compare_superpoint(out: dict, ref: dict, desc_tol: float = 1e-3, score_tol: float = 1e-5):
    """out/ref: dicts with keypoints (N,2), scores (N,), descriptors (256,N) (CPU tensors).

    Keypoints are compared as SETS of integer pixel coordinates (raw top-k order is not even
    stable between two precisions of the reference); any keypoint present on one
    side only must be explained by a near-tie at the selection boundary.  Descriptors and
    scores are compared on the common keypoints."""
    ka = {(int(x), int(y)): i for i, (x, y) in enumerate(out["keypoints"].tolist())}
    kb = {(int(x), int(y)): i for i, (x, y) in enumerate(ref["keypoints"].tolist())}
    common = sorted(set(ka) & set(kb))
    only_a, only_b = set(ka) - set(kb), set(kb) - set(ka)
    ia = torch.tensor([ka[c] for c in common], dtype=torch.long)
    ib = torch.tensor([kb[c] for c in common], dtype=torch.long)
    ds = (out["scores"][ia] - ref["scores"][ib]).abs().max().item() if common else 0.0
    dd = (out["descriptors"][:, ia] - ref["descriptors"][:, ib]).abs().max().item() if common else 0.0
    res = {"n_out": len(ka), "n_ref": len(kb), "common": len(common), "only_out": len(only_a), "only_ref": len(only_b),
           "max_score_diff": ds, "max_desc_diff": dd}
    # boundary explanation: a one-sided keypoint must have a score within score_tol of the
    # weakest selected score on the other side (the top-k cut) or of the threshold.
    if only_a or only_b:
        cut = min(out["scores"].min().item(), ref["scores"].min().item())
        for c in only_a:
            assert abs(out["scores"][ka[c]].item() - cut) <= 10 * score_tol, (c, res)
        for c in only_b:
            assert abs(ref["scores"][kb[c]].item() - cut) <= 10 * score_tol, (c, res)
    assert len(ka) == len(kb), res
    assert ds <= score_tol, res
    assert dd <= desc_tol, res
    return res


def order_is_reference_like(out: dict, k_limited: bool):
    """Score-descending when top-k was applied, row-major (y, x) otherwise (SPN:74-78,183-186)."""
    s = out["scores"]
    if k_limited:
        assert bool((s[:-1] >= s[1:]).all())
    else:
        k = out["keypoints"]
        lin = k[:, 1] * 100000 + k[:, 0]
        assert bool((lin[:-1] < lin[1:]).all())


def compare_lightglue(out: dict, ref: dict, score_tol: float = 1e-3, dense_ref=None, dense_out=None, dense_tol: float = 1e-3,
                      tie_tol: float = 1e-4, max_ties: int = 2, filter_threshold=None):
    """out/ref: reference-style dicts for one pair (CPU tensors; ref from the oracle or the golden
    file).  Integer outputs (stop, prune, matches) must be identical except where the oracle's own
    decision is a numerical near-tie (two assignment scores closer than tie_tol), which is reported:
    that exception needs the oracle's dense log-assignment (dense_ref) — without it any difference fails —
    and is bounded (at most max_ties matches, each checked by match_list_difference_is_a_tie)."""
    res = {}
    assert int(out["stop"]) == int(ref["stop"]), (int(out["stop"]), int(ref["stop"]))
    for k in ("prune0", "prune1"):
        assert torch.equal(out[k].reshape(-1).long(), torch.as_tensor(ref[k]).reshape(-1).long()), k
    a0, b0 = out["matches0"].reshape(-1).long(), torch.as_tensor(ref["matches0"]).reshape(-1).long()
    a1, b1 = out["matches1"].reshape(-1).long(), torch.as_tensor(ref["matches1"]).reshape(-1).long()
    res["n_matches0_mismatch"], res["n_matches1_mismatch"] = int((a0 != b0).sum()), int((a1 != b1).sum())
    mo = out["matches"][0] if isinstance(out["matches"], (list, tuple)) else out["matches"]
    mr = ref["matches"][0] if isinstance(ref["matches"], (list, tuple)) else torch.as_tensor(ref["matches"])
    mo, mr = mo.long().cpu(), mr.long()
    so = out["scores"][0] if isinstance(out["scores"], (list, tuple)) else out["scores"]
    sr = ref["scores"][0] if isinstance(ref["scores"], (list, tuple)) else torch.as_tensor(ref["scores"])
    same = res["n_matches0_mismatch"] == 0 and res["n_matches1_mismatch"] == 0
    if same:
        assert torch.equal(mo, mr), "compact match list"
        common0, common1 = torch.ones_like(a0, dtype=torch.bool), torch.ones_like(a1, dtype=torch.bool)
        if so.numel():
            assert (so.cpu() - sr).abs().max().item() <= score_tol
    else:
        assert dense_ref is not None and dense_ref.numel(), ("integer outputs differ and no log-assignment was given to explain it",
                                                             (a0 != b0).nonzero().reshape(-1).tolist()[:10], (a1 != b1).nonzero().reshape(-1).tolist()[:10])
        # the two sides' mutual matches as (i, j) sets, each side consistent in itself (matches0 and matches1 say the same)
        pairs_of = lambda m0: torch.stack([(m0 >= 0).nonzero().reshape(-1), m0[m0 >= 0]], 1)
        po, pr = pairs_of(a0), pairs_of(b0)
        assert {(int(j), int(i)) for i, j in po.tolist()} == {(int(j), int(i)) for j, i in pairs_of(a1).tolist()}, "matches0 / matches1 disagree"
        ties = match_list_difference_is_a_tie(po, pr, dense_ref, 0.0 if filter_threshold is None else filter_threshold, tie_tol)
        assert len(ties) <= 2 * max_ties, ties
        res["explained_near_ties"] = ties
        touched0 = {t["match"][0] for t in ties}; touched1 = {t["match"][1] for t in ties}
        common0 = torch.tensor([i not in touched0 for i in range(a0.numel())], dtype=torch.bool)
        common1 = torch.tensor([j not in touched1 for j in range(a1.numel())], dtype=torch.bool)
        assert torch.equal(a0[common0], b0[common0]) and torch.equal(a1[common1], b1[common1])
        assert {tuple(x) for x in mo.tolist()} ^ {tuple(x) for x in mr.tolist()} <= {tuple(t["match"]) for t in ties}, "compact match list"
    for k, common in (("matching_scores0", common0), ("matching_scores1", common1)):
        x, y = out[k].reshape(-1), torch.as_tensor(ref[k]).reshape(-1)
        d = (x[common] - y[common]).abs().max().item() if int(common.sum()) else 0.0
        res["max_" + k + "_diff"] = d
        assert d <= score_tol, (k, d)
    if dense_ref is not None and dense_ref.numel():
        m, n = dense_ref.shape[0] - 1, dense_ref.shape[1] - 1
        d = (dense_out[:m, :n] - dense_ref[:m, :n]).abs().max().item()
        res["max_log_assignment_diff"] = d
        assert d <= dense_tol, d
    return res


def match_list_difference_is_a_tie(got: torch.Tensor, want: torch.Tensor, log_assignment: torch.Tensor, filter_threshold: float = 0.0,
                                   tie_tol: float = 1e-4, ind0=None, ind1=None):
    """The near-tie rule of compare_lightglue's docstring, for compact match lists: every match that only one side reports must
    be within tie_tol — IN THE ORACLE'S OWN log-assignment — of being the mutual best of its row and column (or of the filter
    threshold), and the decision it lost / won must itself be that close (top-2 margin of the row or column <= tie_tol).
    Returns the list of explained differences (empty when the lists are equal); raises AssertionError on an unexplained one.
    (LightGlue's dense log-assignment carries ~3e-4 of fp32 noise at 2048 x 2048 against an fp64 evaluation — DESIGN.md section 4,
    yardstick test — so a margin below 1e-4 is not decidable in fp32 by ANY implementation, the reference's included.)"""
    g = {tuple(int(v) for v in x) for x in got.tolist()}
    w = {tuple(int(v) for v in x) for x in want.tolist()}
    la = log_assignment
    # with point pruning the oracle's log-assignment lives in the PRUNED index space: ind0 / ind1 (its taps) list the surviving
    # keypoints; a match on a pruned keypoint cannot be explained
    pos0 = {int(v): k for k, v in enumerate(ind0.tolist())} if ind0 is not None else None
    pos1 = {int(v): k for k, v in enumerate(ind1.tolist())} if ind1 is not None else None
    explained = []
    for (gi, gj) in sorted(g ^ w):
        assert (pos0 is None or gi in pos0) and (pos1 is None or gj in pos1), ("match on a keypoint the oracle pruned", (gi, gj))
        i, j = (pos0[gi] if pos0 is not None else gi), (pos1[gj] if pos1 is not None else gj)
        row, col = la[i, :-1], la[:-1, j]
        v = float(la[i, j])
        near_best = float(row.max()) - v <= tie_tol and float(col.max()) - v <= tie_tol
        top2r, top2c = torch.topk(row, min(2, row.numel())).values, torch.topk(col, min(2, col.numel())).values
        row_margin = float(top2r[0] - top2r[-1]) if row.numel() > 1 else float("inf")
        col_margin = float(top2c[0] - top2c[-1]) if col.numel() > 1 else float("inf")
        near_thr = abs(float(torch.exp(la[i, j])) - filter_threshold) <= tie_tol
        ok = near_best and (min(row_margin, col_margin) <= tie_tol or near_thr)
        assert ok, ("unexplained match difference", (gi, gj), "only_out" if (gi, gj) in g else "only_ref", v, row_margin, col_margin)
        explained.append({"match": (gi, gj), "side": "only_out" if (gi, gj) in g else "only_ref", "log_assignment": v,
                          "row_top2_margin": row_margin, "col_top2_margin": col_margin})
    return explained
