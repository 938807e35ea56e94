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
