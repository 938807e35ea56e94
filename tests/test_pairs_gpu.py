"""GPU: matching_lowres pair generation at the reference's sizes (resize_max 1000, 2048 keypoints, 7 layers)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref, tile_ref
from tests.parity import compare_lightglue, compare_superpoint, order_is_reference_like

pytestmark = pytest.mark.gpu
pairs_mod = importlib.import_module("deep-image-matching_amd.pairs")
weights = importlib.import_module("deep-image-matching_amd.weights")


def test_lowres_pairs_batched_equals_one_call_per_pair(hip_lib):
    rng = np.random.default_rng(0)
    base = (rng.random((1200, 1600)) * 255).astype(np.float32)
    images = [base, np.roll(base, 40, axis=1).copy(), (rng.random((1000, 1500)) * 255).astype(np.float32), base[:, ::-1].copy(),
              (rng.random((1600, 1200)) * 255).astype(np.float32)]
    names = [f"im{i}.jpg" for i in range(5)]
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(0), weights.synthetic_lightglue_state_dict(0, 256)
    a = pairs_mod.LowresPairSelector(sp_sd, lg_sd, pair_batch=8, lib=hip_lib)
    b = pairs_mod.LowresPairSelector(sp_sd, lg_sd, pair_batch=1, lib=hip_lib)
    ta = a.extract(images)
    assert ta[0].shape == (5, 2048, 2) and int(ta[2].min()) == 2048 and float(ta[3].max()) <= 1000.0
    idx = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    ca, cb = a.match_counts(ta, idx), b.match_counts(b.extract(images), idx)
    assert ca.shape == (10,) and np.array_equal(ca, cb)
    sel = a.select(names, images)
    assert sel == [(names[i], names[j]) for (i, j), c in zip(idx, ca) if c > 20]


def _cpu(res):
    return {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in res.items()}


def test_lowres_chain_vs_the_oracle_at_the_reference_sizes(hip_lib):
    """VERDICT r2 next #1b: pairs_from_lowres (pairs_generator.py:103-146) at its real sizes — resize_max 1000, hloc's SuperPoint
    wrapper (nms 3 / 2048 kpts / thr 0.0005 / fix_sampling), LightGlue with 7 layers, depth 0.9 / width 0.95, keypoint-extent
    image size — HIP chain vs ORACLE chain on hardware, stage by stage: resize bit-exact, SuperPoint keypoint sets equal and
    descriptors <= 1e-3, LightGlue matches / stop / prune identical on the same features, and the selected pair list equal.
    4 images: two large related ones, one portrait, one 600 x 800 image that the reference ENLARGES (INTER_AREA's bilinear
    emulation).  Run at the reference's filter_threshold (0.3) and at 0 so the compared match lists are not empty."""
    rng = np.random.default_rng(0)
    base = (rng.random((1200, 1600)) * 255).astype(np.float32)
    images = [base, np.roll(base, 40, axis=1).copy(), (rng.random((1600, 1100)) * 255).astype(np.float32),
              (rng.random((600, 800)) * 255).astype(np.float32)]
    sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(0), weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    sel = pairs_mod.LowresPairSelector(sp_sd, lg_sd, pair_batch=4, lib=hip_lib)
    kt, dt, nt, st = sel.extract(images)
    feats = []
    for i, im in enumerate(images):
        _, _, new = tile_ref.preselection_sizes(im.shape, 1000)
        small = tile_ref.resize_area(im, new) / np.float32(255.0)
        assert max(small.shape) == 1000 and np.array_equal(sel.downsample(im).cpu().numpy(), small)             # incl. the enlargement
        ref = superpoint_ref.superpoint_forward(torch.from_numpy(small)[None, None], sp_sd, dict(pairs_mod.LOWRES_SP_CONF))
        k = int(nt[i])
        kp, sc, de, n = sel._sp.extract_batch_guarded(sel.downsample(im)[None].contiguous())     # the same call extract() made, with the scores
        assert int(n[0]) == k and torch.equal(kp[0, :k], kt[i, :k]) and torch.equal(de[0, :k], dt[i, :k])
        out = {"keypoints": kt[i, :k].cpu(), "scores": sc[0, :k].cpu(), "descriptors": dt[i, :k].t().cpu()}
        res = compare_superpoint(out, ref)
        order_is_reference_like(out, k_limited=True)
        assert res["n_out"] == 2048, res
        kk = kt[i, :k].cpu()
        assert torch.equal(st[i].cpu(), 1 + kk.max(0).values - kk.min(0).values)                                   # LGN:26-27 extent
        feats.append(out)
    idx = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    lgm = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    total = 0
    for th in (pairs_mod.LOWRES_LG_CONF["filter_threshold"], 0.0):
        conf = dict(pairs_mod.LOWRES_LG_CONF, filter_threshold=th)
        mat = lgm.LightGlueHIP(lg_sd, conf, max_pairs=4, max_kpts=2048, lib=hip_lib)
        counts = []
        for s in range(0, len(idx), 4):
            chunk = idx[s:s + 4]
            pidx = torch.tensor(chunk, dtype=torch.int32, device="cuda")
            o = mat.match_batch_guarded(kt, dt, nt, st, pair_idx=pidx, n_pairs=len(chunk))
            o = {k: v.cpu() for k, v in o.items()}
            for j, (a, b) in enumerate(chunk):
                na, nb = int(nt[a]), int(nt[b])
                ref = lightglue_ref.lightglue_forward(feats[a]["keypoints"], feats[a]["descriptors"].t().contiguous(), st[a].cpu(),
                                                      feats[b]["keypoints"], feats[b]["descriptors"].t().contiguous(), st[b].cpu(), lg_sd, conf)
                S = int(o["n_matches"][j])
                assert int(o["stop"][j]) == ref["stop"], (th, a, b)
                assert torch.equal(o["matches"][j, :S], ref["matches"]), (th, a, b)
                assert torch.equal(o["prune01"][j, 0, :na].long(), ref["prune0"].long()) and torch.equal(o["prune01"][j, 1, :nb].long(), ref["prune1"].long())
                if S:
                    assert (o["scores"][j, :S] - ref["scores"]).abs().max().item() <= 1e-3
                counts.append(S)
        if th == pairs_mod.LOWRES_LG_CONF["filter_threshold"]:
            got = sel.match_counts((kt, dt, nt, st), idx)
            assert got.tolist() == counts                                    # the selector's own count path == the oracle's lengths
        total += sum(counts)
    assert total > 0
