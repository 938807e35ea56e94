"""CPU (build container only): the MI355X plugins under the REFERENCE's real base classes.

tests/refstubs.py makes ``deep_image_matching.extractors.extractor_base`` / ``matchers.matcher_base`` importable (the
reference files are executed unmodified from /root/reference; only the absent third-party packages are stood in for), the
plugin module is then imported with ``HAVE_DIM = True`` — its classes subclass the reference's ``ExtractorBase`` /
``MatcherBase`` — and the reference's own control flow is driven with the emulator-built library:

    Config(args)  ->  SuperPointExtractor(config).extract(path)  ->  save_features_h5  ->
    LightGlueMatcher(config).match(features.h5, matches.h5, img0, img1)  ->  raw_matches.h5 / matches.h5

both on full images and through the tiling path (tile_size (400, 300) like the reference's tests/test_pipelines.py:26-30),
plus the MRO with both mixins, the TypeError on a non-Config argument, and the "CUDA out of memory" tile fallback
(matcher_base.py:251-256).  Skipped where /root/reference does not exist (the GPU box)."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from tests import refstubs

pytestmark = pytest.mark.skipif(not refstubs.available(), reason="/root/reference not present")


def _write_images(folder: Path, n=3, hw=(120, 160)):
    from PIL import Image

    folder.mkdir(parents=True)
    rng = np.random.default_rng(0)
    base = (rng.random((hw[0] + 40, hw[1] + 40)) * 255).astype(np.uint8)
    paths = []
    for i in range(n):
        im = base[8 * i: 8 * i + hw[0], 8 * i: 8 * i + hw[1]]   # shifts by whole 8-px cells: the crops share keypoints
        p = folder / f"img{i}.png"
        Image.fromarray(im).save(p)   # single band PNG: rasterio's read() gives (1, H, W)
        paths.append(p)
    return paths


@pytest.fixture
def dim(emu_install, tmp_path):
    """(plugins module bound to the reference base classes, reference Config class, image paths, project dir)."""
    added = refstubs.install(find_fundamental=lambda p0, p1, *a: (np.eye(3), np.ones((len(p0), 1), np.uint8)))
    for k in [k for k in sys.modules if k.startswith("deep-image-matching_amd.plugins")]:
        del sys.modules[k]
    try:
        plugins = importlib.import_module("deep-image-matching_amd.plugins")
        assert plugins.HAVE_DIM, "the reference base classes did not import"
        config = importlib.import_module("deep_image_matching.config")
        imgs = _write_images(tmp_path / "images")
        yield plugins, config, imgs, tmp_path
    finally:
        refstubs.uninstall(added)
        for k in [k for k in sys.modules if k.startswith("deep-image-matching_amd.plugins")]:
            del sys.modules[k]
        importlib.import_module("deep-image-matching_amd.plugins")


def _config(config_mod, project: Path, tiling="none", extra_general=None):
    yml = project / f"user_{tiling}.yaml"
    general = {"geom_verification": "NONE", "min_inliers_per_pair": 1, "min_inlier_ratio_per_pair": 0.0, "tile_size": [80, 64], "tile_overlap": 0,
               "allow_synthetic_weights": True}
    general.update(extra_general or {})
    import yaml
    weights = importlib.import_module("deep-image-matching_amd.weights")
    # checkpoints on disk in the official key layout, as a user would supply them (weights_path)
    torch.save(weights.synthetic_superpoint_state_dict(1234), project / "sp.pth")
    torch.save(weights.synthetic_lightglue_state_dict(0, 256, n_layers=2, gain=1.0), project / "lg.pth")
    yml.write_text(yaml.safe_dump({
        "general": general,
        "extractor": {"name": "superpoint", "max_keypoints": 400, "nms_radius": 2, "keypoint_threshold": 0.001, "remove_borders": 2,
                      "weights_path": str(project / "sp.pth")},
        "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0,
                    "weights_path": str(project / "lg.pth"), "pruning_min_kpts": -1}}))
    return config_mod.Config({"dir": str(project), "pipeline": "superpoint+lightglue", "strategy": "bruteforce", "tiling": tiling,
                              "force": True, "config_file": str(yml), "outs": str(project / f"out_{tiling}")})


def test_plugins_subclass_the_reference_base_classes(dim):
    plugins, config_mod, imgs, project = dim
    eb = importlib.import_module("deep_image_matching.extractors.extractor_base")
    mb = importlib.import_module("deep_image_matching.matchers.matcher_base")
    assert issubclass(plugins.SuperPointExtractor, eb.ExtractorBase) and issubclass(plugins.AlikedExtractor, eb.ExtractorBase)
    assert issubclass(plugins.LightGlueMatcher, mb.MatcherBase)
    # the batched mixins come first in the MRO: their _extract_by_tile / _match_by_tile override the base class's loops
    mro = plugins.SuperPointExtractor.__mro__
    assert mro.index(importlib.import_module("deep-image-matching_amd.tiling").BatchedTilingMixin) < mro.index(eb.ExtractorBase)
    mro = plugins.LightGlueMatcher.__mro__
    assert mro.index(importlib.import_module("deep-image-matching_amd.tile_matching").BatchedTileMatchingMixin) < mro.index(mb.MatcherBase)
    with pytest.raises(TypeError):   # extractor_base.py:127-130 / matcher_base.py:103-106
        plugins.SuperPointExtractor({"general": {}, "extractor": {}})
    with pytest.raises(TypeError):
        plugins.LightGlueMatcher({"general": {}, "matcher": {}})


def test_extract_save_match_through_the_reference_flow(dim):
    from oracle import lightglue_ref, superpoint_ref

    plugins, config_mod, imgs, project = dim
    cfg = _config(config_mod, project, "none")
    ex = plugins.SuperPointExtractor(cfg)
    feature_path = None
    for p in imgs:
        feature_path = ex.extract(p)          # ExtractorBase.extract: rasterio read -> _extract -> save_features_h5
    assert feature_path == cfg.general["output_dir"] / "features.h5" and feature_path.exists()
    h5 = importlib.import_module("deep_image_matching.io.h5")
    feats = [h5.get_features(feature_path, p.name) for p in imgs]
    from PIL import Image
    for p, f in zip(imgs, feats):
        img = np.asarray(Image.open(p)).astype(np.float32)
        ref = superpoint_ref.superpoint_forward(torch.tensor(img / 255.0)[None, None], ex._sd, ex._net_cfg)
        assert f["keypoints"].shape == (400, 2) and f["descriptors"].shape == (256, 400) and f["image_size"].tolist() == [120, 160]
        assert f["keypoints"].dtype == np.float32 and np.all(f["tile_idx"] == 0)
        # features.h5 stores float16 (Q6): integer pixel coordinates < 2048 survive exactly
        assert set(map(tuple, f["keypoints"].astype(int).tolist())) == set(map(tuple, ref["keypoints"].long().tolist()))
    m = plugins.LightGlueMatcher(cfg, local_features="superpoint")
    matches_path = cfg.general["output_dir"] / "matches.h5"
    out = m.match(feature_path, matches_path, imgs[0], imgs[1])      # MatcherBase.match: h5 read -> _match_pairs -> h5 writes
    sz = torch.tensor([120.0, 160.0])
    k0, d0 = torch.from_numpy(feats[0]["keypoints"]), torch.from_numpy(feats[0]["descriptors"].T.copy())
    k1, d1 = torch.from_numpy(feats[1]["keypoints"]), torch.from_numpy(feats[1]["descriptors"].T.copy())
    ref = lightglue_ref.lightglue_forward(k0, d0, sz, k1, d1, sz, m._sd, {**m._conf})
    assert out is not None and out.dtype == np.int64 and np.array_equal(out, ref["matches"].numpy()) and len(out) >= 8
    raw = h5.get_matches(cfg.general["output_dir"] / "raw_matches.h5", imgs[0].name, imgs[1].name)
    ver = h5.get_matches(matches_path, imgs[0].name, imgs[1].name)
    assert np.array_equal(np.asarray(raw), out) and np.array_equal(np.asarray(ver), out)


def test_tiling_path_and_oom_fallback_through_the_reference_flow(dim):
    plugins, config_mod, imgs, project = dim
    cfg = _config(config_mod, project, "grid")
    constants = importlib.import_module("deep_image_matching.constants")
    assert cfg.general["tile_selection"] == constants.TileSelection.GRID and cfg.general["tile_size"] == (80, 64)
    ex = plugins.SuperPointExtractor(cfg)
    for p in imgs[:2]:
        fp = ex.extract(p)                   # -> BatchedTilingMixin._extract_by_tile (2 x 2 tiles of 80 x 64 px ... (160, 120))
    h5 = importlib.import_module("deep_image_matching.io.h5")
    f0, f1 = h5.get_features(fp, imgs[0].name), h5.get_features(fp, imgs[1].name)
    assert set(np.unique(f0["tile_idx"]).astype(int)) == {0, 1, 2, 3} and f0["keypoints"].shape[0] > 400
    # the REAL ExtractorBase._extract_by_tile (EB:279-390: Tiler, one _extract per tile, the numpy shift / border filter / np.unique
    # merge) on the same array == the batched override with its on-device merge (csrc/tile_merge.hip), bit for bit
    base = importlib.import_module("deep_image_matching.extractors.extractor_base").ExtractorBase
    rng = np.random.default_rng(5)
    for shape, ov in (((130, 150), 16),):                            # padded tiles + overlapping tiles (duplicates)
        img = rng.integers(0, 256, shape).astype(np.float32)
        ex.config["general"]["tile_overlap"] = ov
        want = base._extract_by_tile(ex, img.copy(), select_unique=True)
        got = ex._extract_by_tile(img.copy(), select_unique=True)
        assert want["keypoints"].shape[0] > 100
        for k in ("keypoints", "descriptors", "scores", "tile_idx"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (k, shape, ov)
    ex.config["general"]["tile_overlap"] = cfg.general["tile_overlap"]
    m = plugins.LightGlueMatcher(cfg, local_features="superpoint")
    matches_path = cfg.general["output_dir"] / "matches.h5"
    by_tile = m.match(fp, matches_path, imgs[0], imgs[1])            # -> BatchedTileMatchingMixin._match_by_tile (GRID)
    assert by_tile is not None and by_tile.shape[1] == 2
    tm = importlib.import_module("deep-image-matching_amd.tile_matching")
    expect = tm.match_tile_pairs_batched(m._ensure_pairs, f0, f1, [(0, 0), (1, 1), (2, 2), (3, 3)], "cpu")
    assert np.array_equal(by_tile, expect)
    # try_full_image + an out-of-memory error from the library -> the reference falls back to tiles (matcher_base.py:244-256)
    capi = importlib.import_module("deep-image-matching_amd.capi")
    calls = {"n": 0}
    orig = m._match_pairs

    def oom_once(a, b):
        calls["n"] += 1
        if calls["n"] == 1:
            raise capi.DimHipError("CUDA out of memory (HIP): hipMalloc of 123 bytes failed: out of memory")
        return orig(a, b)

    m._match_pairs = oom_once
    (cfg.general["output_dir"] / "raw_matches.h5").unlink()
    matches_path.unlink()
    fb = m.match(fp, matches_path, imgs[0], imgs[1], try_full_image=True)
    assert calls["n"] == 1 and np.array_equal(fb, expect)
    m._match_pairs = lambda a, b: (_ for _ in ()).throw(RuntimeError("some other failure"))
    with pytest.raises(RuntimeError, match="some other failure"):
        m.match(fp, matches_path, imgs[0], imgs[1], try_full_image=True)
