"""CPU (emulator): csrc/sort_ops.hip — the integer bookkeeping of tile-wise matching on the device — against the numpy / torch statements it
replaces: get_features_by_tile's boolean masks (matcher_base.py:1380-1391) and _match_by_tile's np.unique(matches, axis=0) (:452-459)."""
import ctypes

import numpy as np
import pytest
import torch


def p(t):
    return ctypes.c_void_p(t.data_ptr())


def _group(lib, tile_idx, kp, de, row_of_tile, cap):
    n, D, T = int(tile_idx.shape[0]), int(de.shape[1]), int(row_of_tile.shape[0])
    rows = int(row_of_tile.max()) + 1
    kt, dt = torch.zeros(rows, cap, 2), torch.zeros(rows, cap, D)
    it, nt = torch.zeros(rows, cap, dtype=torch.int64), torch.zeros(rows, dtype=torch.int32)
    lib.dim_op_group_by_tile_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(int(lib.dim_op_group_by_tile_workspace_bytes(n, T)), dtype=torch.uint8)
    assert lib.dim_op_group_by_tile(p(tile_idx), int(tile_idx.stride(0)), p(kp), int(kp.stride(0)), p(de), int(de.stride(0)), n, D, p(row_of_tile), T, cap, p(kt), p(dt), p(it), p(nt), p(ws), None) == 0, lib.dim_last_error()
    return kt, dt, it, nt


@pytest.mark.parametrize("n,T", [(300, 6), (4096, 16), (9001, 12)])      # one chunk; exactly one; three (an odd run in the first merge pass)
def test_group_by_tile_is_the_boolean_mask_order(emu_lib, n, T):
    g = torch.Generator().manual_seed(n)
    packed = torch.randn(n, 4 + 8, generator=g)                               # the tiled pipeline's layout: [keypoint 2 | score | tile_idx | descriptor D] per row
    packed[:, 3] = torch.randint(0, T, (n,), generator=g).float()
    tile_idx, kp, de = packed[:, 3], packed[:, 0:2], packed[:, 4:]            # strided views
    counts = torch.zeros(T, dtype=torch.int32)
    assert emu_lib.dim_op_tile_counts(p(tile_idx), int(tile_idx.stride(0)), n, T, p(counts), None) == 0
    assert counts.tolist() == [int((tile_idx == t).sum()) for t in range(T)]
    used = [t for t in range(T) if t % 3 != 1]                               # some tiles are not needed
    row_of_tile = torch.full((T,), -1, dtype=torch.int32)
    row_of_tile[torch.tensor(used)] = torch.arange(len(used), dtype=torch.int32)
    cap = int(counts.max())
    kt, dt, it, nt = _group(emu_lib, tile_idx, kp, de, row_of_tile, cap)
    for r, t in enumerate(used):
        idx = (tile_idx == t).nonzero()[:, 0]                                # the mask order of get_features_by_tile
        c = len(idx)
        assert int(nt[r]) == c and torch.equal(it[r, :c], idx)
        assert torch.equal(kt[r, :c], kp[idx]) and torch.equal(dt[r, :c], de[idx])
        assert not kt[r, c:].any() and not dt[r, c:].any()


def _unique(lib, keys, n_slots, cap_m, i64=False):
    n = int(keys.numel())
    rows = torch.full((n_slots, cap_m, 2), -7, dtype=torch.int64 if i64 else torch.int32)
    cnt, full = torch.full((n_slots,), -1, dtype=torch.int32), torch.full((n_slots,), -1, dtype=torch.int32)
    lib.dim_op_unique_match_rows_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(int(lib.dim_op_unique_match_rows_workspace_bytes(ctypes.c_longlong(n))), dtype=torch.uint8)
    assert lib.dim_op_unique_match_rows(p(keys), ctypes.c_longlong(n), n_slots, cap_m, int(i64), p(rows), p(cnt), p(full), p(ws), None) == 0, lib.dim_last_error()
    return rows, cnt, full


@pytest.mark.parametrize("n_live,n_dead,n_slots", [(50, 10, 1), (4096, 0, 3), (5000, 7000, 4), (13000, 100, 5), (0, 64, 2)])
def test_unique_match_rows_is_np_unique_per_image_pair(emu_lib, n_live, n_dead, n_slots):
    rng = np.random.default_rng(n_live + n_slots)
    slot = rng.integers(0, n_slots, n_live)
    a, b = rng.integers(0, 40, n_live), rng.integers(0, 50, n_live)          # small ranges: many duplicates
    if n_live > 100:
        a[: n_live // 2] = rng.integers(0, 900000, n_live // 2)              # and large indices (20 bits each)
    keys = (slot.astype(np.uint64) << np.uint64(40)) | (a.astype(np.uint64) << np.uint64(20)) | b.astype(np.uint64)
    allk = np.concatenate([keys, np.full(n_dead, np.uint64(0xFFFFFFFFFFFFFFFF))])
    rng.shuffle(allk)
    t = torch.from_numpy(allk.view(np.int64).copy())
    want = [np.unique(np.stack([a[slot == s], b[slot == s]], 1), axis=0) if (slot == s).any() else np.zeros((0, 2), np.int64) for s in range(n_slots)]
    cap_m = max(1, max(len(w) for w in want))
    for i64 in (False, True):
        rows, cnt, full = _unique(emu_lib, t, n_slots, cap_m, i64)
        for s in range(n_slots):
            assert int(cnt[s]) == int(full[s]) == len(want[s])
            assert np.array_equal(rows[s, : len(want[s])].numpy(), want[s])
            assert (rows[s, len(want[s]):] == -7).all()                      # nothing written past the count
    if cap_m > 3:                                                            # an explicit cap below the real count: the first rows, and the full count is reported
        rows, cnt, full = _unique(emu_lib, t, n_slots, cap_m - 3)
        for s in range(n_slots):
            assert int(full[s]) == len(want[s]) and int(cnt[s]) == min(len(want[s]), cap_m - 3)
            assert np.array_equal(rows[s, : int(cnt[s])].numpy(), want[s][: int(cnt[s])])


def test_tile_match_keys(emu_lib):
    g = torch.Generator().manual_seed(3)
    T, cap, NK, b = 5, 30, 12, 4
    it = torch.randint(0, 1 << 19, (T, cap), generator=g, dtype=torch.int64)
    matches = torch.randint(0, cap, (b, NK, 2), generator=g, dtype=torch.int64)
    nm = torch.tensor([0, 5, 12, 7], dtype=torch.int32)
    pidx = torch.tensor([[0, 3], [1, 4], [2, 2], [4, 0]], dtype=torch.int32)
    slot = torch.tensor([9, 9, 2, 1000], dtype=torch.int32)
    keys = torch.zeros(b, NK, dtype=torch.int64)
    assert emu_lib.dim_op_tile_match_keys(p(matches), p(nm), p(it), p(pidx), p(slot), b, NK, cap, p(keys), None) == 0
    for j in range(b):
        for k in range(NK):
            if k < int(nm[j]):
                want = (int(slot[j]) << 40) | (int(it[pidx[j, 0], matches[j, k, 0]]) << 20) | int(it[pidx[j, 1], matches[j, k, 1]])
                assert int(keys[j, k]) == want
            else:
                assert int(keys[j, k]) == -1
