"""CPU: tiling glue — known answers of the reference's own tests/test_tiling.py, and the batched
tile extraction == per-tile _extract + the reference's merge rules (emulator build)."""
import importlib

import numpy as np
import pytest
import torch

tiling = importlib.import_module("deep-image-matching_amd.tiling")
plugins = importlib.import_module("deep-image-matching_amd.plugins")


def test_reference_known_answers():
    img = np.random.RandomState(0).randint(0, 255, (100, 100, 3)).astype(np.uint8)
    tiles, origins, pad = tiling.compute_tiles_by_size(img, 50, 0)       # tests/test_tiling.py:22-54
    assert len(tiles) == 4 and len(origins) == 4 and pad == (0, 0, 0, 0) and all(t.shape == (50, 50, 3) for t in tiles.values())
    tiles, origins, pad = tiling.compute_tiles_by_size(img, 40, 0)       # tests/test_tiling.py:57-88
    assert len(tiles) == 9 and pad == (10, 10, 10, 10) and all(t.shape == (40, 40, 3) for t in tiles.values())
    assert origins[0] == (-10, -10) and origins[8] == (70, 70)
    tiles, origins, pad = tiling.compute_tiles_by_size(img, 50, 10)      # tests/test_tiling.py:91-121
    assert len(tiles) == 4 and pad == (0, 0, 0, 0) and origins[3] == (40, 40)
    tiles, origins, pad = tiling.compute_tiles_by_size(img[..., 0], (50, 25), 0)  # (x, y) window on a 2-D image
    assert len(tiles) == 8 and tiles[0].shape == (25, 50, 1)
    with pytest.raises(TypeError):
        tiling.compute_tiles_by_size(img, "40")
    # 6000 x 4000 with (1500, 1000) tiles -> the 4 x 4 = 16 tiles of BASELINE config 5, no padding
    pad = tiling.compute_padding((4000, 6000), (1000, 1500))
    assert pad == (0, 0, 0, 0)


def test_batched_tile_extraction_equals_sequential(emu_install):
    cfg = {"general": {"tile_size": (40, 32), "tile_overlap": 8},
           "extractor": {"name": "superpoint", "nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 20, "remove_borders": 2, "allow_synthetic_weights": True}}
    ex = plugins.SuperPointExtractor(cfg)
    ex.tile_batch = 4
    img = (torch.rand(60, 70, generator=torch.Generator().manual_seed(2)) * 255).round().numpy().astype(np.float32)
    feats = ex._extract_by_tile(img)
    # sequential reference flow: one _extract per tile + the same merge
    tiles, origins, _ = tiling.compute_tiles_by_size(img, cfg["general"]["tile_size"], cfg["general"]["tile_overlap"])
    per_tile = {i: ex._extract(t[..., 0]) for i, t in tiles.items()}
    ref = tiling.merge_tile_features(per_tile, origins, img.shape, 256)
    for k in ("keypoints", "descriptors", "scores", "tile_idx"):
        assert np.array_equal(feats[k], ref[k]), k
    assert feats["keypoints"].shape[0] > 0 and feats["descriptors"].shape[0] == 256
    kp = feats["keypoints"]
    assert (kp[:, 0] >= 2).all() and (kp[:, 0] < 70 - 2).all() and (kp[:, 1] >= 2).all() and (kp[:, 1] < 60 - 2).all()
    assert (np.lexsort((kp[:, 1], kp[:, 0])) == np.arange(len(kp))).all()  # np.unique order (x, then y)


@pytest.mark.parametrize("model,dim", [("aliked-n16rot", 128), ("aliked-t16", 64)])
def test_batched_tile_extraction_equals_sequential_aliked(emu_install, model, dim):
    """the same for ALIKED (RGB tiles; 128- and 64-wide descriptor rows through gather_tiles / merge_tiles)"""
    cfg = {"general": {"tile_size": (48, 40), "tile_overlap": 8},
           "extractor": {"name": "aliked", "model_name": model, "max_num_keypoints": 30, "nms_radius": 2, "allow_synthetic_weights": True}}
    ex = plugins.AlikedExtractor(cfg)
    ex.tile_batch = 4
    img = (torch.rand(70, 90, 3, generator=torch.Generator().manual_seed(3)) * 255).round().numpy().astype(np.float32)
    feats = ex._extract_by_tile(img)
    tiles, origins, _ = tiling.compute_tiles_by_size(img, cfg["general"]["tile_size"], cfg["general"]["tile_overlap"])
    per_tile = {i: ex._extract(t) for i, t in tiles.items()}
    ref = tiling.merge_tile_features(per_tile, origins, img.shape, dim)
    for k in ("keypoints", "descriptors", "scores", "tile_idx"):
        assert np.array_equal(feats[k], ref[k]), k
    assert feats["keypoints"].shape[0] > 20 and feats["descriptors"].shape[0] == dim
