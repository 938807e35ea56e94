"""GPU (MI355X): the on-device tail of _extract_by_tile (csrc/tile_merge.hip) at config-5 size — 16 tiles x 4000 keypoints x 128-d —
against the numpy statement of EB:330-390, bit for bit, through the C ABI."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
tiling = importlib.import_module("deep-image-matching_amd.tiling")


@pytest.mark.parametrize("overlap", [0, 100])
def test_device_merge_at_config5_size(hip_lib, overlap):
    T, cap, D, H, W = 16, 4000, 128, 4000, 6000
    rng = np.random.default_rng(overlap)
    n = rng.integers(3000, cap + 1, T).astype(np.int32)
    n[5] = cap
    # sub-pixel keypoints inside a 1500 x 1000 tile; with overlap, half-pixel-quantised ones in the shared strips collide across tiles
    kp = (rng.random((T, cap, 2)) * np.array([1500, 1000])).astype(np.float32)
    if overlap:
        kp[:, ::3] = np.round(kp[:, ::3] / 8) * 8
    sc = rng.random((T, cap)).astype(np.float32)
    de = rng.standard_normal((T, cap, D)).astype(np.float32)
    st = (1500 - overlap, 1000 - overlap)
    origins = [(c * st[0], r * st[1]) for r in range(4) for c in range(4)]
    per_tile = {t: {"keypoints": kp[t, :n[t]].copy(), "scores": sc[t, :n[t]].copy(), "descriptors": de[t, :n[t]].T.copy()} for t in range(T)}
    ref = tiling.merge_tile_features(per_tile, dict(enumerate(origins)), (H, W), D, True)
    dev = torch.device("cuda")
    tables = [tuple(torch.from_numpy(a).to(dev) for a in (kp, sc, de, n))]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    got = tiling.merge_tile_features_device(hip_lib, dev, stream, tables, origins, list(range(T)), (H, W), True)
    assert ref["keypoints"].shape[0] > 40000
    if overlap:
        assert ref["keypoints"].shape[0] < int(n.sum()) - 50          # real duplicates were removed
    for k in ("keypoints", "descriptors", "scores", "tile_idx"):
        assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k
    same = tiling.merge_tile_features_device(hip_lib, dev, stream, tables, origins, list(range(T)), (H, W), False)
    cat = tiling.merge_tile_features(per_tile, dict(enumerate(origins)), (H, W), D, False)
    for k in cat:
        assert np.array_equal(same[k], cat[k]), k
