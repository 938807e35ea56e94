"""CPU: the oracle (oracle/*.py) must reproduce the committed golden vectors, which are the
REFERENCE modules' own outputs recorded by oracle/make_golden.py."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref
from tests import golden_cases as gc

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("name", list(gc.SP_CASES))
def test_superpoint_oracle_matches_reference_golden(name):
    case = gc.SP_CASES[name]
    g = np.load(GOLD / f"sp_{name}.npz")
    out = superpoint_ref.superpoint_forward(gc.sp_image(case), gc.sp_weights(case), case["cfg"], taps=True)
    # bit-exact for the index work, fp32 rounding for the float work
    assert np.array_equal(out["keypoints"].numpy(), g["keypoints"])
    assert np.array_equal(out["scores"].numpy(), g["scores"])
    np.testing.assert_allclose(out["descriptors"].numpy(), g["descriptors"], atol=1e-6)
    np.testing.assert_allclose(out["score_map"][0].numpy(), g["score_map"], atol=1e-7)
    if case["cfg"]["max_keypoints"] >= 0:
        assert out["keypoints"].shape[0] <= case["cfg"]["max_keypoints"]


@pytest.mark.parametrize("name", list(gc.LG_CASES))
def test_lightglue_oracle_matches_reference_golden(name):
    case = gc.LG_CASES[name]
    g = np.load(GOLD / f"lg_{name}.npz")
    f0, f1 = gc.lg_inputs(case)
    out = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"],
                                          gc.lg_weights(case), case["conf"], taps=True)
    assert out["stop"] == int(g["stop"])
    # no vacuous goldens: every case except the "no keypoints" exit holds matches (and so exercises matches / scores)
    assert name == "prune_to_empty" or g["matches"].shape[0] > 0
    assert np.array_equal(out["matches0"].numpy(), g["matches0"])
    assert np.array_equal(out["matches1"].numpy(), g["matches1"])
    assert np.array_equal(out["matches"].numpy(), g["matches"])
    assert np.array_equal(out["prune0"].long().numpy(), g["prune0"].astype(np.int64))
    assert np.array_equal(out["prune1"].long().numpy(), g["prune1"].astype(np.int64))
    np.testing.assert_allclose(out["matching_scores0"].numpy(), g["matching_scores0"], atol=1e-5)
    np.testing.assert_allclose(out["scores"].numpy(), g["scores"], atol=1e-5)


def test_simple_nms_known_answers():
    """SURVEY Appendix D KATs taken from the reference's simple_nms (r = 1)."""
    row = torch.tensor([[0.1, 0.9, 0.5, 0.4, 0.1, 0.1, 0.1, 0.1, 0.1]])
    got = superpoint_ref.simple_nms(row[None], 1)[0, 0]
    assert torch.equal(got, torch.tensor([0, 0.9, 0, 0.4, 0, 0.1, 0.1, 0.1, 0.1]))
    tie = torch.tensor([[0.2, 0.9, 0.9, 0.2]])
    got = superpoint_ref.simple_nms(tie[None], 1)[0, 0]
    assert got[1] == 0.9 and got[2] == 0.9


def test_filter_matches_first_max_tie_rule():
    s = torch.full((3, 5), -5.0)
    s[0, 1] = s[0, 2] = -1.0  # tie in row 0 -> first index (1)
    s[1, 1] = -0.5
    m0, m1, ms0, ms1 = lightglue_ref.filter_matches(s, 0.0)
    assert m0.tolist()[0] in (-1, 1)
    assert m0.shape == (2,) and m1.shape == (4,)
