"""CPU: the ExtractorBase/MatcherBase-shaped plugin hooks (_extract / _match_pairs) driven through
the emulator-built library, checked against the oracle; wrapper-level error behaviour."""
import importlib

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref

plugins = importlib.import_module("deep-image-matching_amd.plugins")
weights = importlib.import_module("deep-image-matching_amd.weights")


def test_superpoint_extractor_hook_contract(emu_lib):
    cfg = {"general": {}, "extractor": {"name": "superpoint", "nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 40}}
    ex = plugins.SuperPointExtractor(cfg, _lib=emu_lib, _device="cpu")
    assert ex.grayscale and ex.descriptor_size == 256 and ex.required_inputs == ["image"]
    img = (torch.rand(48, 64, generator=torch.Generator().manual_seed(4)) * 255).numpy().astype(np.float32)  # 0..255 as EB feeds it
    f = ex._extract(img)
    assert set(f) == {"keypoints", "scores", "descriptors"}
    assert f["keypoints"].dtype == np.float32 and f["keypoints"].shape == (40, 2)
    assert f["descriptors"].shape == (256, 40) and f["scores"].shape == (40,)
    ref = superpoint_ref.superpoint_forward(torch.tensor(img / 255.0, dtype=torch.float)[None, None], ex._sd, ex._net_cfg)
    assert set(map(tuple, f["keypoints"].astype(int).tolist())) == set(map(tuple, ref["keypoints"].long().tolist()))
    # a larger image re-sizes the resident handle transparently
    f2 = ex._extract(np.zeros((56, 72), np.float32))
    assert f2["keypoints"].shape[1] == 2


def test_lightglue_matcher_hook_contract(emu_lib):
    cfg = {"general": {}, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}}
    m = plugins.LightGlueMatcher(cfg, local_features="superpoint", _lib=emu_lib, _device="cpu")
    assert m.min_matches == 20 and m.max_feat_no_tiling == 200000
    g = torch.Generator().manual_seed(1)
    k0, k1 = torch.rand(20, 2, generator=g) * 100, torch.rand(28, 2, generator=g) * 100
    d0 = torch.nn.functional.normalize(torch.randn(20, 256, generator=g), dim=-1)
    d1 = torch.nn.functional.normalize(torch.randn(28, 256, generator=g), dim=-1)
    size = np.array([100, 120], np.int32)  # (H, W) as DIM stores it
    f0 = {"keypoints": k0.numpy(), "descriptors": d0.t().numpy(), "scores": np.ones(20, np.float32), "tile_idx": np.zeros(20, np.float32), "image_size": size}
    f1 = {"keypoints": k1.numpy(), "descriptors": d1.t().numpy(), "scores": np.ones(28, np.float32), "image_size": size}
    out = m._match_pairs(f0, f1)
    assert out.dtype == np.int64 and out.ndim == 2 and out.shape[1] == 2
    sz = torch.tensor([100.0, 120.0])
    ref = lightglue_ref.lightglue_forward(k0, d0, sz, k1, d1, sz, m._sd, {**m._conf})
    assert np.array_equal(out, ref["matches"].numpy())
    with pytest.raises(KeyError):
        m._match_pairs({"keypoints": k0.numpy()}, f1)
    with pytest.raises(ValueError):
        m._match_pairs({"keypoints": k0.numpy(), "descriptors": np.zeros((7, 9), np.float32)}, f1)
    # empty side -> (0, 2) result, like the reference's "no keypoints" exit
    e = {"keypoints": np.zeros((0, 2), np.float32), "descriptors": np.zeros((0, 256), np.float32), "image_size": size}
    assert m._match_pairs(e, f1).shape == (0, 2)


def test_aliked_extractor_hook_contract(emu_lib):
    from oracle import aliked_ref

    cfg = {"general": {}, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 50, "nms_radius": 2}}
    ex = plugins.AlikedExtractor(cfg, _lib=emu_lib, _device="cpu")
    assert not ex.grayscale and ex.descriptor_size == 128
    img = (torch.rand(48, 64, 3, generator=torch.Generator().manual_seed(6)) * 255).numpy().astype(np.float32)  # HxWx3 RGB 0..255
    f = ex._extract(img)
    assert set(f) == {"keypoints", "scores", "descriptors"}
    assert f["keypoints"].shape == (50, 2) and f["descriptors"].shape == (128, 50) and f["scores"].shape == (50,)
    ref = aliked_ref.aliked_forward(torch.tensor(img.transpose(2, 0, 1)[None] / 255.0, dtype=torch.float), ex._sd, ex._net_cfg)
    a = {tuple(np.round(k).astype(int)) for k in f["keypoints"]}
    b = {tuple(np.round(k).astype(int)) for k in ref["keypoints"].numpy()}
    assert len(a ^ b) <= 2


def test_plugin_arithmetic_option_switches_the_library_mode(emu_lib):
    cfg = {"general": {}, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1,
                                      "filter_threshold": 0.0, "arithmetic": "bf16x6"}}
    try:
        m = plugins.LightGlueMatcher(cfg, _lib=emu_lib, _device="cpu")
        g = torch.Generator().manual_seed(1)
        k0, k1 = torch.rand(20, 2, generator=g) * 100, torch.rand(28, 2, generator=g) * 100
        d0 = torch.nn.functional.normalize(torch.randn(20, 256, generator=g), dim=-1)
        d1 = torch.nn.functional.normalize(torch.randn(28, 256, generator=g), dim=-1)
        size = np.array([100, 120], np.int32)
        f0 = {"keypoints": k0.numpy(), "descriptors": d0.t().numpy(), "image_size": size}
        f1 = {"keypoints": k1.numpy(), "descriptors": d1.t().numpy(), "image_size": size}
        a = m._match_pairs(f0, f1)
        emu_lib.dim_tune_set(1, 2)
        b = m._match_pairs(f0, f1)
        assert np.array_equal(a, b)  # both modes are fp32-class: same matches
        with pytest.raises(ValueError):
            plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", "arithmetic": "fp8"}}, _lib=emu_lib, _device="cpu")
    finally:
        emu_lib.dim_tune_set(1, 2)
