"""CPU: the comparators themselves — the near-tie rule accepts only what the oracle's own log-assignment margins explain."""
import pytest
import torch

from tests.parity import compare_lightglue


def _side(m0, n):
    m0 = torch.tensor(m0)
    m1 = torch.full((n,), -1)
    for i, j in enumerate(m0.tolist()):
        if j >= 0:
            m1[j] = i
    pairs = torch.tensor([[i, j] for i, j in enumerate(m0.tolist()) if j >= 0])
    return {"stop": 3, "prune0": torch.zeros(len(m0)), "prune1": torch.zeros(n), "matches0": m0, "matches1": m1,
            "matching_scores0": torch.where(m0 >= 0, torch.tensor(0.36), torch.tensor(0.0)),
            "matching_scores1": torch.where(m1 >= 0, torch.tensor(0.36), torch.tensor(0.0)),
            "matches": [pairs], "scores": [torch.full((len(pairs),), 0.36)]}


def test_near_tie_rule_of_compare_lightglue():
    m, n = 6, 5
    la = torch.full((m + 1, n + 1), -20.0)
    for i, j in ((0, 1), (2, 3), (4, 0)):
        la[i, j] = -1.0
    la[2, 4] = -1.00005                                   # row 2: two candidates 5e-5 apart in the ORACLE's own scores
    ref, tie, real = _side([1, -1, 3, -1, 0, -1], n), _side([1, -1, 4, -1, 0, -1], n), _side([1, -1, 2, -1, 0, -1], n)
    assert compare_lightglue(ref, ref)["n_matches0_mismatch"] == 0
    res = compare_lightglue(tie, ref, dense_ref=la, dense_out=la)
    assert res["n_matches0_mismatch"] == 1 and len(res["explained_near_ties"]) == 2
    with pytest.raises(AssertionError, match="unexplained match difference"):
        compare_lightglue(real, ref, dense_ref=la, dense_out=la)       # (2, 2) scores -20 in the oracle: not a tie
    with pytest.raises(AssertionError, match="no log-assignment"):
        compare_lightglue(tie, ref)                                     # without the oracle's scores nothing is excused
    la[2, 4] = -1.01                                                    # 1e-2 apart: a real disagreement
    with pytest.raises(AssertionError, match="unexplained match difference"):
        compare_lightglue(tie, ref, dense_ref=la, dense_out=la)
