"""CPU (emulator build): randomised property tests of the small integer / index kernels (hypothesis): the device tile merge equals
the numpy statement of EB:330-390 on arbitrary ragged tables, simple_nms is idempotent on its own output's support and agrees with
the oracle, packed match rows round-trip."""
import ctypes
import importlib

import numpy as np
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import superpoint_ref

tiling = importlib.import_module("deep-image-matching_amd.tiling")
p = lambda t: ctypes.c_void_p(t.data_ptr())
SET = dict(max_examples=25, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), T=st.integers(1, 6), cap=st.integers(1, 70), D=st.sampled_from([1, 7, 128, 130]),
       grid=st.integers(2, 40), unique=st.booleans())
def test_device_tile_merge_equals_numpy_on_random_tables(emu_lib, seed, T, cap, D, grid, unique):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(10, 120)), int(rng.integers(10, 160))
    n = rng.integers(0, cap + 1, T).astype(np.int32)
    kp = (rng.integers(0, grid, (T, cap, 2)) * float(rng.choice([0.5, 1.0, 1.25]))).astype(np.float32)   # coarse grid: many duplicates
    sc = rng.random((T, cap)).astype(np.float32)
    de = rng.standard_normal((T, cap, D)).astype(np.float32)
    origins = [(int(rng.integers(-12, W)), int(rng.integers(-12, H))) for _ in range(T)]
    ids = sorted(rng.choice(50, T, replace=False).tolist())
    per_tile = {ids[t]: {"keypoints": kp[t, :n[t]].copy(), "scores": sc[t, :n[t]].copy(), "descriptors": de[t, :n[t]].T.copy()} for t in range(T)}
    ref = tiling.merge_tile_features(per_tile, {ids[t]: origins[t] for t in range(T)}, (H, W), D, unique)
    tables = [(torch.from_numpy(kp), torch.from_numpy(sc), torch.from_numpy(de), torch.from_numpy(n))]
    got = tiling.merge_tile_features_device(emu_lib, torch.device("cpu"), None, tables, origins, ids, (H, W), unique)
    for k in ("keypoints", "descriptors", "scores", "tile_idx"):
        assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), H=st.integers(9, 70), W=st.integers(9, 90), radius=st.integers(1, 4), levels=st.sampled_from([0, 4, 64]))
def test_simple_nms_equals_oracle_and_is_stable(emu_lib, seed, H, W, radius, levels):
    """levels > 0: scores quantised to a few values -> plateaus and exact ties, the case the == comparisons of SPN:47-63 decide."""
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(1, H, W, generator=g)
    if levels:
        s = (s * levels).floor() / levels
    out = torch.empty_like(s)
    assert emu_lib.dim_op_simple_nms_f32(p(s), p(out), 1, H, W, radius, None) == 0, emu_lib.dim_last_error()
    ref = superpoint_ref.simple_nms(s, radius)
    assert torch.equal(out, ref)
    # survivors keep their score, everything else is exactly 0; applying it again to its own output changes nothing
    assert bool(((out == 0) | (out == s)).all())
    again = torch.empty_like(s)
    assert emu_lib.dim_op_simple_nms_f32(p(out.contiguous()), p(again), 1, H, W, radius, None) == 0
    assert torch.equal(again, superpoint_ref.simple_nms(out, radius))


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), P=st.integers(1, 5), NK=st.integers(1, 40))
def test_match_rows_round_trip(emu_lib, seed, P, NK):
    g = torch.Generator().manual_seed(seed)
    cnt = torch.randint(0, NK + 1, (P,), generator=g, dtype=torch.int32)
    m = torch.randint(0, 5000, (P, NK, 2), generator=g, dtype=torch.int64)
    sc = torch.rand(P, NK, generator=g)
    rows = torch.full((P, NK, 3), -1, dtype=torch.int32)
    assert emu_lib.dim_op_pack_match_rows(p(m), p(sc), p(cnt), NK, P, p(rows), None) == 0, emu_lib.dim_last_error()
    for q in range(P):
        k = int(cnt[q])
        assert torch.equal(rows[q, :k, :2].long(), m[q, :k])
        assert torch.equal(rows[q, :k, 2].view(torch.float32), sc[q, :k])
        assert bool((rows[q, k:] == 0).all())
