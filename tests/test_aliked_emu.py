"""CPU: the HIP ALIKED sources on the test emulator vs the oracle (which is pinned bit-exactly
against the reference's aliked.py by oracle/make_golden.py) and vs the reference golden."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import aliked_ref
from tests import golden_cases as gc

al_mod = importlib.import_module("deep-image-matching_amd.aliked_hip")
GOLD = Path(__file__).parent / "golden"


def compare_aliked(out, ref, kp_tol=1e-3, desc_tol=1e-3, score_tol=1e-3, max_missing=0, label=None, ref_score_map=None, threshold=0.2,
                   tie_tol=2e-5, nms_radius=3, n_limit=None):
    """north_star's bar: keypoint set exact, descriptors / scores within 1e-3 (keypoint coordinates likewise: they are
    sub-pixel soft-argmax outputs in pixels).  Every measured maximum is returned and, with ``label``, appended to
    gpurun_out/parity_measured.jsonl so that the numbers behind the assertion are on record."""
    # keypoints are sub-pixel (NMS pixel + soft-argmax offset, which can exceed half a pixel): pair each output keypoint with the
    # NEAREST reference keypoint (rounding to the pixel grid would split a pair whose offset sits at x.5) and require a bijection
    from scipy.spatial import cKDTree
    ko, kr = out["keypoints"].numpy().astype(np.float64), ref["keypoints"].numpy().astype(np.float64)
    pairs = []
    if len(ko) and len(kr):
        dist, nn = cKDTree(kr).query(ko)
        used = set()
        for i, (d, j) in enumerate(zip(dist, nn)):
            if d <= 0.05 and int(j) not in used:       # 0.05 px: far below the NMS spacing, far above any rounding difference
                used.add(int(j)); pairs.append((i, int(j)))
    res = {"n_out": len(ko), "n_ref": len(kr), "common": len(pairs)}
    ia = torch.tensor([p[0] for p in pairs], dtype=torch.long); ib = torch.tensor([p[1] for p in pairs], dtype=torch.long)
    only_out = sorted(set(range(len(ko))) - {p[0] for p in pairs}); only_ref = sorted(set(range(len(kr))) - {p[1] for p in pairs})
    res["kp"] = (out["keypoints"][ia] - ref["keypoints"][ib]).abs().max().item() if pairs else 0.0       # (an image without detections: nothing to compare)
    res["score"] = (out["scores"][ia] - ref["scores"][ib]).abs().max().item() if pairs else 0.0
    res["desc"] = (out["descriptors"][:, ia] - ref["descriptors"][:, ib]).abs().max().item() if pairs else 0.0
    if label is not None:
        import json
        d = Path(__file__).resolve().parents[1] / "gpurun_out"
        try:
            d.mkdir(exist_ok=True)
            with open(d / "parity_measured.jsonl", "a") as f:
                f.write(json.dumps({"case": label, **res}) + "\n")
        except OSError:
            pass
    if (only_out or only_ref) and ref_score_map is not None:
        # The keypoint SET must be exact except for numerical near-ties of the selection itself: DKD keeps the n_limit highest
        # NMS maxima above the threshold (ALN:170-186), so a keypoint found on one side only must sit — in the REFERENCE's own
        # score map — within tie_tol of the weakest selected score (the n_limit cut), of the threshold, or of another pixel of
        # its own NMS window (an exact-equality tie of simple_nms).  (The reference is
        # not stable there either: its fp32 and fp64 evaluations of tests/golden case gray_limit differ in 5 of 60 keypoints.)
        sm = ref_score_map.reshape(ref_score_map.shape[-2], ref_score_map.shape[-1])
        # the NMS pixel of a refined keypoint is the local maximum within one pixel of its rounded position
        def nms_score(xy):
            x, y = int(round(xy[0])), int(round(xy[1]))
            return float(sm[max(0, y - 1): y + 2, max(0, x - 1): x + 2].max())
        def nms_tie(xy, radius=nms_radius):
            """a second pixel inside the NMS window within tie_tol of the maximum: simple_nms compares floats with == (ALN:66-89), so
            on a plateau (the sigmoid saturates near 1) the reference keeps BOTH pixels while a 1e-7 difference keeps one"""
            x, y = int(round(xy[0])), int(round(xy[1]))
            win = sm[max(0, y - radius): y + radius + 1, max(0, x - radius): x + radius + 1].reshape(-1)
            top = torch.topk(win, 2).values
            return float(top[0] - top[1])
        # the n_limit cut of the REFERENCE: the n_limit-th largest NMS maximum above the threshold (DKD, ALN:150-186)
        nms = aliked_ref._simple_nms(ref_score_map.reshape(1, 1, *sm.shape), nms_radius)[0, 0].clone()
        nms[:nms_radius] = 0; nms[-nms_radius:] = 0; nms[:, :nms_radius] = 0; nms[:, -nms_radius:] = 0
        cand = torch.sort(sm[nms > threshold], descending=True).values
        cut = float(cand[n_limit - 1]) if n_limit is not None and len(cand) > n_limit else threshold
        res["one_sided"] = {"out": [(ko[i].tolist(), nms_score(ko[i]), nms_tie(ko[i])) for i in only_out],
                            "ref": [(kr[j].tolist(), nms_score(kr[j]), nms_tie(kr[j])) for j in only_ref], "cut": cut}
        for xy in [ko[i] for i in only_out] + [kr[j] for j in only_ref]:
            v = nms_score(xy)
            assert min(abs(v - cut), abs(v - threshold), nms_tie(xy)) <= tie_tol, (tuple(xy), v, cut, nms_tie(xy), res)
        res["near_tie_keypoints"] = max(len(only_out), len(only_ref))
        max_missing = max(max_missing, res["near_tie_keypoints"])
    assert len(ko) == len(kr) and len(pairs) >= len(kr) - max_missing, res
    assert res["kp"] <= kp_tol and res["score"] <= score_tol and res["desc"] <= desc_tol, res
    return res


@pytest.mark.parametrize("name", list(gc.AL_CASES))
def test_aliked_emulated_vs_oracle_and_golden(emu_lib, name):
    case = gc.AL_CASES[name]
    sd, img = gc.al_weights(case), gc.al_image(case)
    net = al_mod.AlikedHIP(sd, case["cfg"], max_batch=1, max_hw=(case["H"], case["W"]), capacity=4096, device="cpu", lib=emu_lib)
    out = {k: v.cpu() for k, v in net(img).items()}
    ref = aliked_ref.aliked_forward(img, sd, case["cfg"], taps=True)
    taps = net.debug_taps()
    pt, pl = taps["pad"]
    H, W = case["H"], case["W"]
    fm = torch.nn.functional.normalize(taps["x1234"][0, pt:pt + H, pl:pl + W].permute(2, 0, 1), dim=0)
    assert (fm - ref["feature_map"][0]).abs().max().item() < 1e-3
    res = compare_aliked(out, ref)
    g = np.load(GOLD / f"al_{name}.npz")
    gold = {k: torch.from_numpy(g[k]) for k in ("keypoints", "scores", "descriptors")}
    compare_aliked(out, gold)


def test_aliked_top_k_mode_vs_oracle(emu_lib):
    """detection_threshold <= 0 = DKD's top-k mode (ALN:602, 150-151): the max_num_keypoints highest NMS maxima, whatever their score."""
    case = gc.AL_CASES["rgb_pad"]
    cfg = {**case["cfg"], "detection_threshold": -1.0, "max_num_keypoints": 50}
    sd, img = gc.al_weights(case), gc.al_image(case)
    net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(case["H"], case["W"]), capacity=64, device="cpu", lib=emu_lib)
    out = {k: v.cpu() for k, v in net(img).items()}
    ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
    assert ref["keypoints"].shape[0] == 50
    res = compare_aliked(out, ref, ref_score_map=ref["score_map"], threshold=0.0, nms_radius=cfg["nms_radius"], n_limit=50)
    assert res["n_out"] == 50
    # with the default threshold the same image keeps 336 keypoints (golden rgb_pad): the 50 are its highest maxima, none below 0.2 is needed
    thr = {**cfg, "detection_threshold": 0.2}
    net2 = al_mod.AlikedHIP(sd, thr, max_batch=1, max_hw=(case["H"], case["W"]), capacity=64, device="cpu", lib=emu_lib)
    out2 = {k: v.cpu() for k, v in net2(img).items()}
    assert {tuple(r) for r in out["keypoints"].round().int().tolist()} == {tuple(r) for r in out2["keypoints"].round().int().tolist()}


def test_aliked_mean_threshold_mode_vs_oracle(emu_lib):
    """detection_threshold <= 0 AND max_num_keypoints <= 0: the reference's top_k is then <= 0 and DKD keeps every NMS maximum above
    the image's MEAN score (ALN:161-163) — at most n_limit_max = 20000 of them (ALN:571), which is the slot size of the keep-all modes."""
    case = gc.AL_CASES["rgb_pad"]
    cfg = {**case["cfg"], "detection_threshold": -1.0, "max_num_keypoints": -1}
    sd, img = gc.al_weights(case), gc.al_image(case)
    net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(case["H"], case["W"]), device="cpu", lib=emu_lib)
    assert net.capacity == aliked_ref.N_LIMIT_MAX == 20000
    out = {k: v.cpu() for k, v in net(img).items()}
    ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
    mean = float(ref["score_map"].mean())
    assert ref["keypoints"].shape[0] > 100
    res = compare_aliked(out, ref, ref_score_map=ref["score_map"], threshold=mean, nms_radius=cfg["nms_radius"])
    assert res["n_out"] == ref["keypoints"].shape[0]


def _fill_reference(img, sd, cfg, ref_full, k):
    """What DKD's top-k mode returns when the image has FEWER NMS maxima than top_k (ALN:150-151), with the zero-score fill pixels fixed to
    the first non-maximum pixels in row-major order (which pixels torch.topk takes among the equal zeros is an accident of its sort: libstdc++'s
    heap / introselect on the CPU, a radix select on CUDA): the reference's arithmetic (oracle dkd_refine + SDDH) at those indices."""
    r = cfg["nms_radius"]
    nms = aliked_ref.dkd_nms_map(ref_full["score_map"], r).reshape(-1)
    maxima = (nms > 0).nonzero()[:, 0]
    order = torch.sort(nms[maxima], descending=True, stable=True)[1]
    fill = (nms <= 0).nonzero()[:k - len(maxima), 0]
    idx = torch.cat([maxima[order], fill])
    return aliked_ref.aliked_forward(img, sd, cfg, taps=True, idx=idx), len(maxima)


def test_aliked_top_k_mode_fills_up_with_zero_score_pixels(emu_lib):
    """DKD's top-k mode on an image with fewer NMS maxima than max_num_keypoints: the reference's torch.topk still returns top_k indices — the
    maxima, score-descending, then zero-score pixels (ALN:150-151).  Count, order of the maxima and the values at every returned pixel must be
    the reference's; the fill pixels' identity is pinned to row-major order (see _fill_reference)."""
    case = gc.AL_CASES["rgb_pad"]
    sd, img = gc.al_weights(case), gc.al_image(case)
    base = {**case["cfg"], "detection_threshold": -1.0}
    ref_all = aliked_ref.aliked_forward(img, sd, {**base, "max_num_keypoints": -1}, taps=True)
    n_max = int((aliked_ref.dkd_nms_map(ref_all["score_map"], base["nms_radius"]) > 0).sum())
    k = n_max + 37
    cfg = {**base, "max_num_keypoints": k}
    ref_torch = aliked_ref.aliked_forward(img, sd, cfg)
    assert ref_torch["keypoints"].shape[0] == k          # the reference's count: top_k, not the number of maxima
    net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(case["H"], case["W"]), device="cpu", lib=emu_lib)
    out = {kk: v.cpu() for kk, v in net(img).items()}
    assert out["keypoints"].shape[0] == k
    ref, m = _fill_reference(img, sd, cfg, ref_all, k)
    assert m == n_max
    # the maxima: the reference's own torch.topk output (paired by position: near-equal scores may swap neighbours in the order)
    part = lambda d, a, b: {kk: d[kk][..., a:b] if kk == "descriptors" else d[kk][a:b] for kk in ("keypoints", "scores", "descriptors")}  # noqa: E731
    res = compare_aliked(part(out, 0, m), part(ref_torch, 0, m))
    assert res["common"] == m
    # the fill, in order
    assert (out["keypoints"][m:] - ref["keypoints"][m:]).abs().max().item() <= 1e-3
    assert (out["scores"][m:] - ref["scores"][m:]).abs().max().item() <= 1e-3
    assert (out["descriptors"][:, m:] - ref["descriptors"][:, m:]).abs().max().item() <= 1e-3
    # score-descending order of the maxima (torch.topk sorts).  The true keypoint scores are not exported (Q8): pair every output keypoint with
    # its twin in `ref` (whose maxima are in stable score-descending order) and read the NMS score there
    from scipy.spatial import cKDTree
    nms = aliked_ref.dkd_nms_map(ref_all["score_map"], base["nms_radius"]).reshape(-1)
    v_sorted = torch.sort(nms[nms > 0], descending=True, stable=True)[0]
    dist, twin = cKDTree(ref["keypoints"][:m].numpy().astype(np.float64)).query(out["keypoints"][:m].numpy().astype(np.float64))
    assert dist.max() <= 0.05 and len(set(twin.tolist())) == m
    v = v_sorted[torch.from_numpy(twin)]
    assert (v[1:] <= v[:-1] + 2e-5).all()


def test_aliked_more_than_4096_keypoints(emu_lib):
    """config/aliked.yaml asks for max_num_keypoints 8000 and ALN:571 allows 20000: above 4096 the selection runs as radix select -> 4096-key
    chunk sorts -> rank merge (sp_post.hip).  128 x 192 noise at nms_radius 1 has ~4400 maxima: k = 4200 takes the select path (n > k), k = 5000
    the sort-everything + zero-fill path (n < k), both across two chunks."""
    case = gc.AL_CASES["rgb_pad"]
    sd = gc.al_weights(case)
    H, W = 128, 192
    img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(5))
    base = {**case["cfg"], "nms_radius": 1, "detection_threshold": -1.0}
    ref_all = aliked_ref.aliked_forward(img, sd, {**base, "max_num_keypoints": -1}, taps=True)
    n_max = int((aliked_ref.dkd_nms_map(ref_all["score_map"], 1) > 0).sum())
    assert 4200 < n_max < 5000, n_max
    for k in (4200, 5000):
        cfg = {**base, "max_num_keypoints": k}
        net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(H, W), device="cpu", lib=emu_lib)
        out = {kk: v.cpu() for kk, v in net(img).items()}
        assert out["keypoints"].shape[0] == k
        if k < n_max:
            ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
            res = compare_aliked(out, ref, ref_score_map=ref["score_map"], threshold=0.0, nms_radius=1, n_limit=k)
            assert res["n_out"] == k and res["common"] >= k - 4, res
        else:
            ref, m = _fill_reference(img, sd, cfg, ref_all, k)
            res = compare_aliked({kk: v[..., :m] if kk == "descriptors" else v[:m] for kk, v in out.items()},
                                 {kk: ref[kk][..., :m] if kk == "descriptors" else ref[kk][:m] for kk in ("keypoints", "scores", "descriptors")},
                                 ref_score_map=ref_all["score_map"], threshold=0.0, nms_radius=1)
            assert res["common"] >= m - 4, res
            if res["common"] == m:   # identical maxima sets: the fill pixels are then the same row-major prefix on both sides
                assert (out["keypoints"][m:] - ref["keypoints"][m:]).abs().max().item() <= 1e-3
                assert (out["descriptors"][:, m:] - ref["descriptors"][:, m:]).abs().max().item() <= 1e-3


REAL_ALIKED = Path(__file__).parent / "assets" / "aliked-n16rot.pth"   # byte copy of the reference's thirdparty/ALIKED/models/aliked-n16rot.pth
REAL_ALIKED_N32 = Path(__file__).parent / "assets" / "aliked-n32.pth"  # likewise (md5 fb7434eaaf6c52604541322d7e0fde58)


@pytest.mark.parametrize("model", ["aliked-n16rot", "aliked-n32", "aliked-t16"])
def test_aliked_real_checkpoint_through_the_hip_sources(emu_lib, model):
    """The REAL aliked-n16rot.pth / aliked-n32.pth that ship inside the reference tree (tests/assets holds byte copies: data files, md5
    bfec5e8086e9f6bf68ffeb90ca7a793a / fb7434eaaf6c52604541322d7e0fde58, so that the GPU tests can use them too) through the HIP sources
    on the emulator, vs the oracle that oracle/make_golden.py pins bit-exact against the reference's aliked.py with the same files.  Real
    weights have the trained dynamic range (BatchNorm scales, score head) that the seeded synthetic ones lack."""
    path = Path(__file__).parent / "assets" / f"{model}.pth"
    if not path.exists():
        pytest.skip(f"{model}.pth asset not present")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.load_aliked_state_dict(str(path), model_name=model)
    cfg = {"model_name": model, "max_num_keypoints": 200, "detection_threshold": 0.2, "nms_radius": 2}
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(96.0), indexing="ij")
    img = (0.5 + 0.25 * torch.sin(xx / 5.0) * torch.cos(yy / 7.0))[None, None].repeat(1, 3, 1, 1) \
        + 0.2 * torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(3))
    img = img.clamp(0, 1)
    net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(64, 96), capacity=4096, device="cpu", lib=emu_lib)
    out = {k: v.cpu() for k, v in net(img).items()}
    ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
    taps = net.debug_taps()
    pt, pl = taps["pad"]
    fm = torch.nn.functional.normalize(taps["x1234"][0, pt:pt + 64, pl:pl + 96].permute(2, 0, 1), dim=0)
    assert (fm - ref["feature_map"][0]).abs().max().item() < 1e-3
    res = compare_aliked(out, ref)
    assert res["n_out"] >= 8   # (the n32 checkpoint finds 9 keypoints on this 64 x 96 pattern, n16rot 37)
