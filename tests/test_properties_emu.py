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


from oracle import tile_ref


def _resize(lib, fn, img, h, w):
    src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    dst = torch.empty(h, w, dtype=torch.float32)
    assert getattr(lib, fn)(p(src), img.shape[0], img.shape[1], p(dst), h, w, 0, None) == 0, lib.dim_last_error()
    return dst.numpy()


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), H=st.integers(2, 70), W=st.integers(2, 70), h=st.integers(1, 90), w=st.integers(1, 90))
def test_resize_kernels_equal_the_restated_opencv_tables(emu_lib, seed, H, W, h, w):
    """dim_op_resize_area_f32 (shrinking and enlarging) and dim_op_resize_linear_f32 on arbitrary size pairs, bit for bit against
    oracle/tile_ref.py's restatement of cv2.INTER_AREA / INTER_LINEAR (utils/image.py:47-65)."""
    img = (np.random.default_rng(seed).random((H, W)) * 255).astype(np.float32)
    assert np.array_equal(_resize(emu_lib, "dim_op_resize_area_f32", img, h, w), tile_ref.resize_area(img, (w, h)))
    assert np.array_equal(_resize(emu_lib, "dim_op_resize_linear_f32", img, h, w), tile_ref.resize_linear(img, (w, h)))


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(0, 60), T0=st.integers(1, 6), T1=st.integers(1, 6), tw=st.integers(4, 40), th=st.integers(4, 40))
def test_tile_pair_votes_equal_the_reference_helpers(emu_lib, seed, n, T0, T1, tw, th):
    rng = np.random.default_rng(seed)
    k0, k1 = (rng.random((80, 2)) * 100).astype(np.float32), (rng.random((70, 2)) * 100).astype(np.float32)
    k0[::5] = np.round(k0[::5])          # points exactly on tile borders: the strict inequalities of points_in_rect decide
    m = np.stack([rng.integers(0, 80, n), rng.integers(0, 70, n)], 1).astype(np.int64).reshape(n, 2)
    s0, s1 = np.float32(rng.choice([1.0, 0.5, 0.37])), np.float32(rng.choice([1.0, 2.0, 0.81]))
    o0 = rng.integers(-5, 180, (T0, 2)).astype(np.int32); o1 = rng.integers(-5, 180, (T1, 2)).astype(np.int32)
    o0[0] = (np.round(k0[0] / s0)).astype(np.int32)   # a tile whose corner coincides with a (scaled) keypoint
    v = torch.full((T0, T1), -7, dtype=torch.int32)
    mt = torch.zeros(max(n, 1), 2, dtype=torch.int64); mt[:n] = torch.from_numpy(m)      # (capacity >= 1 row: the entry point rejects a null table)
    kt0, kt1, nt = torch.from_numpy(k0), torch.from_numpy(k1), torch.tensor([n], dtype=torch.int32)
    ot0, ot1 = torch.from_numpy(o0).contiguous(), torch.from_numpy(o1).contiguous()
    rc = emu_lib.dim_op_tile_pair_votes(p(kt0), p(kt1), p(mt), p(nt), max(n, 1), ctypes.c_float(float(s0)), ctypes.c_float(float(s1)), p(ot0), T0,
                                        p(ot1), T1, tw, th, p(v), None)
    assert rc == 0, emu_lib.dim_last_error()
    a, b = k0[m[:, 0]] / s0, k1[m[:, 1]] / s1
    ref = tile_ref.tile_pair_votes(a, b, {i: tuple(x) for i, x in enumerate(o0)}, {i: tuple(x) for i, x in enumerate(o1)}, (tw, th))
    assert np.array_equal(v.numpy(), ref)
