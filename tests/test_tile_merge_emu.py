"""CPU (emulator build): dim_op_merge_tiles — the device tail of _extract_by_tile (EB:330-390) — against the numpy statement of the
same rules (tiling.merge_tile_features: shift, 2-px border filter, np.unique(axis=0, return_index=True)), bit for bit."""
import importlib

import numpy as np
import pytest
import torch

tiling = importlib.import_module("deep-image-matching_amd.tiling")


def _case(seed, T, cap, D, grid, H, W):
    rng = np.random.default_rng(seed)
    n = rng.integers(0, cap + 1, T).astype(np.int32)
    n[rng.integers(0, T)] = 0            # an empty tile
    n[rng.integers(0, T)] = cap          # a full one
    # keypoints on a coarse grid (plus a few sub-pixel ones): duplicates inside a tile, across overlapping tiles, and on the borders
    kp = (rng.integers(0, grid, (T, cap, 2)) * 1.5).astype(np.float32)
    kp[:, ::7] += rng.random((T, kp[:, ::7].shape[1], 2)).astype(np.float32)
    sc = rng.random((T, cap)).astype(np.float32)
    de = rng.standard_normal((T, cap, D)).astype(np.float32)
    origins = [(int(rng.integers(-8, W - 8)), int(rng.integers(-8, H - 8))) for _ in range(T)]
    if T > 1:
        origins[1] = origins[0]          # two tiles on top of each other: cross-tile duplicates
        kp[1, :cap // 2] = kp[0, :cap // 2]
    return kp, sc, de, n, origins


@pytest.mark.parametrize("unique", [True, False])
@pytest.mark.parametrize("T,cap,D", [(5, 40, 16), (3, 300, 130), (1, 17, 256), (9, 64, 128)])
def test_device_merge_equals_numpy_merge(emu_lib, T, cap, D, unique):
    H, W = 90, 120
    kp, sc, de, n, origins = _case(T * 1000 + cap, T, cap, D, 24, H, W)
    ids = list(range(3, 3 + T))          # tile numbers need not start at 0
    per_tile = {ids[t]: {"keypoints": kp[t, :n[t]].copy(), "scores": sc[t, :n[t]].copy(), "descriptors": de[t, :n[t]].T.copy()} for t in range(T)}
    ref = tiling.merge_tile_features(per_tile, {ids[t]: origins[t] for t in range(T)}, (H, W), D, unique)
    tables = [(torch.from_numpy(kp), torch.from_numpy(sc), torch.from_numpy(de), torch.from_numpy(n))]
    got = tiling.merge_tile_features_device(emu_lib, torch.device("cpu"), None, tables, origins, ids, (H, W), unique)
    assert ref["keypoints"].shape[0] > 0 or T == 1
    for k in ("keypoints", "descriptors", "scores", "tile_idx"):
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert np.array_equal(got[k], ref[k]), k
    if unique and T > 1:
        total = sum(int(x) for x in n)
        assert ref["keypoints"].shape[0] < total    # the case really contains duplicates / dropped rows


def test_device_merge_of_chunked_tables_and_empty_result(emu_lib):
    H, W = 90, 120
    kp, sc, de, n, origins = _case(11, 6, 32, 24, 24, H, W)
    ids = list(range(6))
    per_tile = {t: {"keypoints": kp[t, :n[t]].copy(), "scores": sc[t, :n[t]].copy(), "descriptors": de[t, :n[t]].T.copy()} for t in range(6)}
    ref = tiling.merge_tile_features(per_tile, dict(enumerate(origins)), (H, W), 24, True)
    # two extractor batches, the second with a larger capacity (keep-all regrow): padded to one table
    big = lambda x: np.concatenate([x, np.zeros((x.shape[0], 8) + x.shape[2:], x.dtype)], axis=1)
    tables = [tuple(torch.from_numpy(np.ascontiguousarray(a[:4])) for a in (kp, sc, de, n)),
              (torch.from_numpy(big(kp[4:])), torch.from_numpy(big(sc[4:])), torch.from_numpy(big(de[4:])), torch.from_numpy(n[4:].copy()))]
    got = tiling.merge_tile_features_device(emu_lib, torch.device("cpu"), None, tables, origins, ids, (H, W), True)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    # nothing survives: every keypoint on the border
    z = tiling.merge_tile_features_device(emu_lib, torch.device("cpu"), None, [(torch.zeros(2, 8, 2), torch.ones(2, 8), torch.ones(2, 8, 24), torch.full((2,), 8, dtype=torch.int32))],
                                          [(0, 0), (0, 0)], [0, 1], (H, W), True)
    assert z["keypoints"].shape == (0, 2) and z["descriptors"].shape == (24, 0) and z["scores"].shape == (0,) and z["tile_idx"].shape == (0,)


def test_large_merges_take_the_host_path_with_the_same_result(emu_lib, monkeypatch):
    """the device rank sort is O((tiles x capacity)^2); above tiling.DEVICE_MERGE_MAX_SLOTS the merge runs the reference's numpy
    statements on the host (only the live keypoints).  Same result either way: the threshold is lowered so that this case crosses it."""
    H, W = 90, 120
    kp, sc, de, n, origins = _case(77, 6, 48, 32, 24, H, W)
    ids = list(range(6))
    tables = [(torch.from_numpy(kp), torch.from_numpy(sc), torch.from_numpy(de), torch.from_numpy(n))]
    dev_res = tiling.merge_tile_features_device(emu_lib, torch.device("cpu"), None, tables, origins, ids, (H, W), True)
    monkeypatch.setattr(tiling, "DEVICE_MERGE_MAX_SLOTS", 6 * 48 - 1)
    host_res = tiling.merge_tile_features_device(None, torch.device("cpu"), None, tables, origins, ids, (H, W), True)    # lib=None: the device must not be touched
    for k in dev_res:
        assert host_res[k].dtype == dev_res[k].dtype and np.array_equal(host_res[k], dev_res[k]), k
