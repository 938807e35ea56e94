"""CPU: BASELINE configs[0] on its REAL inputs — the oracle must reproduce the reference modules' recorded outputs on the photographs the
reference ships (tests/assets/config1: byte copies of assets/example_sacre_coeur/images/*.jpg and assets/pytest/images/*.jpg), with
config/superpoint+lightglue.yaml's parameters; ALIKED with the TRAINED aliked-n16rot checkpoint.  (VERDICT r4 next #1.  The goldens are
written by `python oracle/make_golden.py config1`, which asserts oracle == reference on every image and pair; here a sample is re-checked so
that the committed files, the committed JPEG bytes and the oracle cannot drift apart.)"""
import importlib
from itertools import combinations
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import aliked_ref, lightglue_ref, superpoint_ref
from tests import golden_cases as gc
from tests.config1_real import check_pixels, compare_sparse, gold, golden_features, lg_golden, stem

weights = importlib.import_module("deep-image-matching_amd.weights")
ALIKED_CKPT = Path(__file__).parent / "assets" / "aliked-n16rot.pth"


def test_assets_are_the_five_config1_images_and_the_pytest_fixture():
    sizes = [gc.real_rgb(n).shape[:2] for n in gc.SACRE_COEUR]
    assert sizes == [(480, 640), (640, 618), (640, 618), (618, 640), (784, 784)]
    assert [gc.real_rgb(n).shape for n in gc.PYTEST_IMAGES] == [(533, 800, 3)] * 3
    assert gc.config1_pairs() == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]  # pairs_generator.py:37-38


def test_gray_conversion_is_the_q5_formula():
    """cv2.cvtColor(RGB array, COLOR_BGR2GRAY), 8-bit fixed point: the R and B weights end up swapped (extractor_base.py:197-200)."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]]], dtype=np.uint8)
    a = px.astype(np.int64)
    want = ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14)
    assert want.tolist() == [[29, 150, 76, 145]]                                             # pure red -> 29 (cv2 on true BGR would give 76)
    from PIL import Image
    import tempfile, os
    d = tempfile.mkdtemp()
    try:
        gc_dir = gc.REAL_DIR
        Image.fromarray(px).save(os.path.join(d, "t.png"))
        gc.REAL_DIR = Path(d)
        assert gc.real_gray("t.png").tolist() == [[29.0, 150.0, 76.0, 145.0]]
    finally:
        gc.REAL_DIR = gc_dir


@pytest.mark.parametrize("name", gc.SACRE_COEUR[:2] + gc.SACRE_COEUR[4:] + gc.PYTEST_IMAGES[:1])
def test_superpoint_oracle_on_the_real_photographs(name):
    g = gold("sp")
    gray = gc.real_gray(name)
    check_pixels(g, name, gray)
    img = torch.tensor(gray[None][None] / 255.0, dtype=torch.float)
    out = superpoint_ref.superpoint_forward(img, weights.synthetic_superpoint_state_dict(1234), gc.CONFIG1_SP)
    assert out["keypoints"].shape[0] == 2000                              # real photographs: far more than 2000 candidates, top-k binds
    assert np.array_equal(out["keypoints"].numpy(), g[stem(name) + "/keypoints"])
    res = compare_sparse(out, g, name, 256, subpixel=False, score_tol=1e-6, desc_tol=1e-5)
    assert res["common"] == 2000 and res["desc_sub_checked"] == 125


@pytest.mark.parametrize("variant,pairs", [("generic", [(0, 1)]), ("generic_t0", [(0, 4), (2, 3)]), ("matching", [(0, 4), (1, 2)])])
def test_lightglue_oracle_on_the_reference_features_of_the_real_photographs(variant, pairs):
    g = gold("lg")
    center = torch.as_tensor(g["center"])
    sd = weights.synthetic_lightglue_matching_state_dict(0, 256, center=center) if variant == "matching" else weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    conf = dict(gc.CONFIG1_LG, filter_threshold=0.0 if variant == "generic_t0" else 0.1)
    for a, b in pairs:
        fa, fb = golden_features("superpoint", gc.SACRE_COEUR[a]), golden_features("superpoint", gc.SACRE_COEUR[b])
        t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
        out = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                              t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, conf)
        ref = lg_golden(g, f"{variant}/{a}_{b}")
        assert out["stop"] == ref["stop"]
        assert torch.equal(out["matches0"].long(), ref["matches0"]) and torch.equal(out["matches"].long(), ref["matches"])
        assert torch.equal(out["prune0"].long(), ref["prune0"]) and torch.equal(out["prune1"].long(), ref["prune1"])
        assert (out["matching_scores0"] - ref["matching_scores0"]).abs().max().item() < (1e-3 if variant == "matching" else 1e-5)


def test_superpoint_to_lightglue_leg_has_real_match_lists_on_the_dsc_photographs():
    """SuperPoint -> LightGlue on real pixels with real match lists.  The three overlapping DSC photographs, the seeded
    SuperPoint's float16 features, matching-capable LightGlue weights whitened on those descriptors (weights.descriptor_whitening; the matrix
    travels in the golden file): 132 / 180 / 140 reference matches at the default threshold 0.1."""
    g = gold("lg")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 256, sharpness=2.0, center=torch.as_tensor(g["dsc/center"]), whiten=torch.as_tensor(g["dsc/whiten"]))
    tags = [f"dsc/{stem(a)}__{stem(b)}" for a, b in combinations(gc.PYTEST_IMAGES, 2)]
    assert [int(g[t_ + "/matches"].shape[0]) for t_ in tags] == [132, 180, 140]
    # the whitening matrix is reproducible from the golden features (so the fixture is data, not a free parameter)
    c, w = weights.descriptor_whitening(torch.cat([torch.as_tensor(golden_features("superpoint", n)["descriptors"].astype(np.float32)).t() for n in gc.PYTEST_IMAGES]))
    assert (c - torch.as_tensor(g["dsc/center"])).abs().max().item() < 1e-6 and (w - torch.as_tensor(g["dsc/whiten"])).abs().max().item() < 1e-2 * float(w.abs().max())
    na, nb = gc.PYTEST_IMAGES[0], gc.PYTEST_IMAGES[2]
    fa, fb = golden_features("superpoint", na), golden_features("superpoint", nb)
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))  # noqa: E731
    out = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                          t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, dict(gc.CONFIG1_LG))
    ref = lg_golden(g, f"dsc/{stem(na)}__{stem(nb)}")
    assert out["stop"] == ref["stop"] and torch.equal(out["matches"].long(), ref["matches"]) and ref["matches"].shape[0] == 180
    assert (out["scores"] - ref["scores"]).abs().max().item() < 1e-3


@pytest.mark.skipif(not ALIKED_CKPT.exists(), reason="aliked-n16rot.pth asset not present")
@pytest.mark.parametrize("name", [gc.SACRE_COEUR[0], gc.PYTEST_IMAGES[2]])
def test_aliked_oracle_trained_checkpoint_on_the_real_photographs(name):
    g = gold("aliked")
    rgb = gc.real_rgb(name)
    check_pixels(g, name, rgb)
    sd = weights.load_aliked_state_dict(str(ALIKED_CKPT), model_name="aliked-n16rot")
    img = torch.tensor(rgb.astype(np.float32).transpose(2, 0, 1)[None] / 255.0, dtype=torch.float)
    out = aliked_ref.aliked_forward(img, sd, gc.CONFIG1_AL)
    assert out["descriptors"].shape[0] == 128                              # (128, N), as extractors/aliked.py:57-58 hands it on
    res = compare_sparse(out, g, name, 128, subpixel=True, score_tol=1e-5, desc_tol=1e-5, kp_tol=1e-4)
    assert res["n_out"] == res["n_ref"] == res["common"] and res["n_ref"] > 1000, {k: v for k, v in res.items() if not k.startswith("only")}


def test_lightglue_oracle_on_the_trained_aliked_features_real_matches():
    """Trained ALIKED descriptors of overlapping photographs + the matching-capable synthetic LightGlue weights at the reference's default
    threshold 0.1: hundreds of matches per pair (the DSC photographs: 852 / 1336 / 1038)."""
    g = gold("aliked_lg")
    sd = weights.synthetic_lightglue_matching_state_dict(0, 128)
    na, nb = gc.PYTEST_IMAGES[0], gc.PYTEST_IMAGES[1]
    fa, fb = golden_features("aliked", na), golden_features("aliked", nb)
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32))
    out = lightglue_ref.lightglue_forward(t(fa["keypoints"]), t(fa["descriptors"]).t().contiguous(), t(fa["image_size"]),
                                          t(fb["keypoints"]), t(fb["descriptors"]).t().contiguous(), t(fb["image_size"]), sd, dict(gc.CONFIG1_LG))
    ref = lg_golden(g, f"{stem(na)}__{stem(nb)}")
    assert ref["matches"].shape[0] == 852
    assert out["stop"] == ref["stop"] and torch.equal(out["matches"].long(), ref["matches"])
    assert (out["scores"] - ref["scores"]).abs().max().item() < 1e-3
    tags = [f"{stem(a)}__{stem(b)}" for grp in (gc.PYTEST_IMAGES, gc.SACRE_COEUR) for a, b in combinations(grp, 2)]
    counts = [int(g[t_ + "/matches"].shape[0]) for t_ in tags]
    assert len(counts) == 13 and min(counts) >= 100
