"""CPU: the ExtractorBase/MatcherBase-shaped plugin hooks (_extract / _match_pairs) driven through
the emulator-built library, checked against the oracle; wrapper-level error behaviour."""
import importlib

import numpy as np
import pytest
import torch

from oracle import lightglue_ref, superpoint_ref

plugins = importlib.import_module("deep-image-matching_amd.plugins")
weights = importlib.import_module("deep-image-matching_amd.weights")
capi = importlib.import_module("deep-image-matching_amd.capi")


def test_superpoint_extractor_hook_contract(emu_install):
    cfg = {"general": {}, "extractor": {"name": "superpoint", "nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 40, "allow_synthetic_weights": True}}
    ex = plugins.SuperPointExtractor(cfg)
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


def test_lightglue_matcher_hook_contract(emu_install):
    cfg = {"general": {}, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0,
                                      "allow_synthetic_weights": True, "pruning_min_kpts": -1}}
    m = plugins.LightGlueMatcher(cfg, local_features="superpoint")
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


def test_aliked_extractor_hook_contract(emu_install):
    from oracle import aliked_ref

    cfg = {"general": {}, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 50, "nms_radius": 2, "allow_synthetic_weights": True}}
    ex = plugins.AlikedExtractor(cfg)
    assert not ex.grayscale and ex.descriptor_size == 128
    img = (torch.rand(48, 64, 3, generator=torch.Generator().manual_seed(6)) * 255).numpy().astype(np.float32)  # HxWx3 RGB 0..255
    f = ex._extract(img)
    assert set(f) == {"keypoints", "scores", "descriptors"}
    assert f["keypoints"].shape == (50, 2) and f["descriptors"].shape == (128, 50) and f["scores"].shape == (50,)
    ref = aliked_ref.aliked_forward(torch.tensor(img.transpose(2, 0, 1)[None] / 255.0, dtype=torch.float), ex._sd, ex._net_cfg)
    a = {tuple(np.round(k).astype(int)) for k in f["keypoints"]}
    b = {tuple(np.round(k).astype(int)) for k in ref["keypoints"].numpy()}
    assert len(a ^ b) <= 2


@pytest.mark.parametrize("model,dim", [("aliked-t16", 64), ("aliked-n32", 128)])
def test_aliked_extractor_variants_with_the_shipped_checkpoints(emu_install, model, dim):
    """`model_name` selects the geometry (ALN:573-579) and `weights_path` the checkpoint the reference ships (tests/assets: byte copies);
    the plugin's descriptor_size follows the model (64 for aliked-t16)."""
    from pathlib import Path

    from oracle import aliked_ref

    ckpt = Path(__file__).parent / "assets" / f"{model}.pth"
    if not ckpt.exists():
        pytest.skip(f"{model}.pth asset not present")
    cfg = {"general": {}, "extractor": {"name": "aliked", "model_name": model, "max_num_keypoints": 40, "nms_radius": 2, "weights_path": str(ckpt)}}
    ex = plugins.AlikedExtractor(cfg)
    assert ex.descriptor_size == dim
    yy, xx = torch.meshgrid(torch.arange(48.0), torch.arange(64.0), indexing="ij")
    img = ((0.5 + 0.3 * torch.sin(xx / 4.0) * torch.cos(yy / 5.0))[..., None].repeat(1, 1, 3) * 255
           + 30 * torch.rand(48, 64, 3, generator=torch.Generator().manual_seed(8))).clamp(0, 255).numpy().astype(np.float32)
    f = ex._extract(img)
    n = f["keypoints"].shape[0]
    assert n > 5 and f["descriptors"].shape == (dim, n) and f["scores"].shape == (n,)
    ref = aliked_ref.aliked_forward(torch.tensor(img.transpose(2, 0, 1)[None] / 255.0, dtype=torch.float), ex._sd, ex._net_cfg)
    assert ref["keypoints"].shape[0] == n
    a = {tuple(np.round(k).astype(int)) for k in f["keypoints"]}
    b = {tuple(np.round(k).astype(int)) for k in ref["keypoints"].numpy()}
    assert len(a ^ b) <= 2


def test_plugin_arithmetic_option_switches_the_library_mode(emu_install):
    cfg = {"general": {}, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1,
                                      "filter_threshold": 0.0, "arithmetic": "bf16x6", "allow_synthetic_weights": True}}
    try:
        m = plugins.LightGlueMatcher(cfg)
        g = torch.Generator().manual_seed(1)
        k0, k1 = torch.rand(20, 2, generator=g) * 100, torch.rand(28, 2, generator=g) * 100
        d0 = torch.nn.functional.normalize(torch.randn(20, 256, generator=g), dim=-1)
        d1 = torch.nn.functional.normalize(torch.randn(28, 256, generator=g), dim=-1)
        size = np.array([100, 120], np.int32)
        f0 = {"keypoints": k0.numpy(), "descriptors": d0.t().numpy(), "image_size": size}
        f1 = {"keypoints": k1.numpy(), "descriptors": d1.t().numpy(), "image_size": size}
        a = m._match_pairs(f0, f1)
        capi.set_arithmetic(emu_install, "fp16x3")
        b = m._match_pairs(f0, f1)
        assert np.array_equal(a, b)  # both modes are fp32-class: same matches
        with pytest.raises(ValueError):
            plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", "arithmetic": "fp8"}})
    finally:
        capi.set_arithmetic(emu_install, "fp16x3")


def test_plugins_refuse_to_run_without_weights(emu_install, monkeypatch):
    """The reference downloads its checkpoints (SPN:149, LGN:383); without a path the plugins raise instead of
    silently producing features from random weights (synthetic weights are an explicit opt-in)."""
    for var in ("DIM_SUPERPOINT_WEIGHTS", "DIM_LIGHTGLUE_WEIGHTS", "DIM_ALIKED_WEIGHTS"):
        monkeypatch.delenv(var, raising=False)
    with pytest.raises(weights.MissingWeightsError):
        plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint"}})
    with pytest.raises(weights.MissingWeightsError):
        plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue"}})
    with pytest.raises(weights.MissingWeightsError):
        plugins.AlikedExtractor({"general": {}, "extractor": {"name": "aliked"}})


def test_plugins_load_checkpoint_files(emu_install, tmp_path):
    """weights_path: the official key layout saved with torch.save loads unchanged (incl. the legacy LightGlue names)."""
    sp_sd = weights.synthetic_superpoint_state_dict(5)
    torch.save(sp_sd, tmp_path / "sp.pth")
    ex = plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", "weights_path": str(tmp_path / "sp.pth"), "max_keypoints": 30}})
    assert all(torch.equal(ex._sd[k], sp_sd[k]) for k in sp_sd)
    lg_sd = weights.synthetic_lightglue_state_dict(3, 256, n_layers=2)
    legacy = {k.replace("transformers.0.self_attn", "self_attn.0").replace("transformers.1.cross_attn", "cross_attn.1"): v for k, v in lg_sd.items()}
    legacy.pop("confidence_thresholds")
    torch.save(legacy, tmp_path / "lg.pth")
    m = plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", "n_layers": 2, "weights_path": str(tmp_path / "lg.pth")}})
    assert set(m._sd) == set(lg_sd) and all(torch.equal(m._sd[k], lg_sd[k]) for k in lg_sd)
    assert m._conf["pruning_min_kpts"] == 1536  # the reference's GPU value with flash attention (LGN:318-323)


def test_keep_all_mode_never_drops_keypoints(emu_install):
    """with max_keypoints = -1 an image with more candidates than the slot is re-extracted with a larger
    slot, so every keypoint the reference returns is returned."""
    cfg = {"general": {}, "extractor": {"name": "superpoint", "nms_radius": 1, "keypoint_threshold": 0.0, "max_keypoints": -1,
                                        "remove_borders": 1, "allow_synthetic_weights": True}}
    ex = plugins.SuperPointExtractor(cfg)
    img = (torch.rand(48, 64, generator=torch.Generator().manual_seed(9)) * 255).numpy().astype(np.float32)
    ex._ensure(48, 64)
    ex._net = None
    ex._capacity = lambda H, W: max(ex._min_capacity, 64)   # force a slot that is too small for the first call
    f = ex._extract(img)
    ref = superpoint_ref.superpoint_forward(torch.tensor(img / 255.0, dtype=torch.float)[None, None], ex._sd, ex._net_cfg)
    assert ref["keypoints"].shape[0] > 64
    assert f["keypoints"].shape[0] == ref["keypoints"].shape[0]
    assert np.array_equal(f["keypoints"], ref["keypoints"].numpy())  # row-major order, like torch.nonzero


def test_arithmetic_is_a_per_handle_choice(emu_install):
    """Round 5: `dim_handle_tune_set` overrides a process default for ONE handle.  Two matchers side by side — one created with
    arithmetic "fp32", one without — keep their own arithmetic across interleaved calls, the process default stays fp16x3, and the override is visible
    in the results (fp32-MFMA and fp16x3 scores differ in the last bits while every integer output agrees)."""
    lib = emu_install
    lg = __import__("importlib").import_module("deep-image-matching_amd.lightglue_hip")
    sd = weights.synthetic_lightglue_state_dict(3, 256, n_layers=2, gain=2.0)
    conf = {"n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
    a = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=64, device="cpu", lib=lib, arithmetic="fp32")
    b = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=64, device="cpu", lib=lib)
    g = torch.Generator().manual_seed(5)
    kt = torch.rand(2, 64, 2, generator=g) * 100
    dt = torch.nn.functional.normalize(torch.randn(2, 64, 256, generator=g), dim=-1)
    nt = torch.tensor([60, 64], dtype=torch.int32)
    st = torch.full((2, 2), 100.0)
    outs = [(h.match_batch(kt, dt, nt, st, n_pairs=1)) for h in (a, b, a, b)]
    assert capi.get_arithmetic(lib) == 2
    assert torch.equal(outs[0]["mscores01"], outs[2]["mscores01"]) and torch.equal(outs[1]["mscores01"], outs[3]["mscores01"])    # each handle reproduces itself
    assert torch.equal(outs[0]["matches01"], outs[1]["matches01"])                                                                  # same decisions
    assert not torch.equal(outs[0]["mscores01"], outs[1]["mscores01"])                                                              # different arithmetic really ran
    assert (outs[0]["mscores01"] - outs[1]["mscores01"]).abs().max().item() < 1e-5
    # back to the process default: handle a now equals handle b bit for bit
    capi.set_handle_arithmetic(lib, a._h, None)
    assert torch.equal(a.match_batch(kt, dt, nt, st, n_pairs=1)["mscores01"], outs[1]["mscores01"])
    # a key without a per-handle form is refused
    assert lib.dim_handle_tune_set(a._h, 6, 2) != 0 and b"per-handle" in lib.dim_last_error()
