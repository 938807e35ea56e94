"""GPU: exhaustive-pair pipeline (feature table + pair index, batched) == the batch-1 plugin hooks,
and full-size properties that do not need the oracle."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_equals_plugin_hooks(hip_lib):
    sp = importlib.import_module("deep-image-matching_amd.superpoint_hip")
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    pl = importlib.import_module("deep-image-matching_amd.pipeline")
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 300, "remove_borders": 4}
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    imgs = torch.rand(6, 192, 256, generator=torch.Generator().manual_seed(3))
    ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=4, max_hw=(192, 256), capacity=300)
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=4, max_kpts=300)
    pipe = pl.PairMatchingPipeline(ext, mat)
    table = pipe.extract_all(imgs.cuda())
    pairs = pl.exhaustive_pairs(6)
    lists = pipe.to_match_lists(*pipe.match_all(table, pairs))
    ex = plugins.SuperPointExtractor({"general": {}, "extractor": dict(cfg, allow_synthetic_weights=True)})
    ma = plugins.LightGlueMatcher({"general": {}, "matcher": dict(conf, allow_synthetic_weights=True, pruning_min_kpts=-1)})
    feats = []
    for i in range(6):
        f = ex._extract((imgs[i] * 255).numpy().astype(np.float32))
        f["image_size"] = np.array([192, 256], np.int32)
        feats.append(f)
        k = int(table[3][i])
        # (x*255)/255 may differ from x in the last ulp, so compare keypoint sets loosely but counts exactly
        assert f["keypoints"].shape[0] == k
    for p, (a, b) in enumerate(pairs.tolist()):
        m = ma._match_pairs(feats[a], feats[b])
        assert m.dtype == np.int64
        assert abs(m.shape[0] - lists[p][0].shape[0]) <= 2  # image round trip through 0..255 may move a near-tie


def test_full_size_properties(hip_lib):
    """BASELINE sizes (1024^2, 2048 kpts): size-independent properties — determinism, symmetry of
    the mutual-NN result under swapping the two images, all-in-bounds indices, score ordering."""
    sp = importlib.import_module("deep-image-matching_amd.superpoint_hip")
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
    imgs = torch.rand(2, 1024, 1024, generator=torch.Generator().manual_seed(8)).cuda()
    ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=2, max_hw=(1024, 1024), capacity=2048)
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, gain=2.0), conf, max_pairs=2, max_kpts=2048)
    kp, sc, de, n = ext.extract_batch(imgs)
    kp2, sc2, de2, n2 = ext.extract_batch(imgs)
    assert torch.equal(kp, kp2) and torch.equal(de, de2) and torch.equal(n, n2)  # deterministic
    assert n.tolist() == [2048, 2048]
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    assert bool(((kp >= 4) & (kp < 1020)).all())  # remove_borders
    assert float((de.norm(dim=-1) - 1).abs().max()) < 1e-5
    # NMS property: no two keypoints within Chebyshev distance nms_radius of each other
    k0 = kp[0].long()
    d = (k0[:, None, :] - k0[None, :, :]).abs().max(-1).values + torch.eye(2048, device=k0.device, dtype=torch.long) * 99
    assert int(d.min()) > 3
    size = torch.full((2, 2), 1024.0, device="cuda")
    pair_idx = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device="cuda")
    o = mat.match_batch(kp, de, n, size, pair_idx=pair_idx)
    s0, s1 = int(o["n_matches"][0]), int(o["n_matches"][1])
    assert s0 == s1 and s0 > 0
    a = o["matches"][0, :s0]
    b = o["matches"][1, :s1]
    assert set(map(tuple, a.tolist())) == set((j, i) for i, j in b.tolist())  # swapping the images transposes the matches
    assert bool((o["stop"] == 9).all())
    assert int(a.max()) < 2048 and int(a.min()) >= 0
    assert len(set(a[:, 0].tolist())) == s0 and len(set(a[:, 1].tolist())) == s0  # one-to-one
