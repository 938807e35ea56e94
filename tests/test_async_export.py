"""CPU: the asynchronous exporter writes exactly what the synchronous stores write (same containers, same layouts),
off the producer thread; the end-to-end runner on the emulator (extract -> match -> verify -> export)."""
import importlib
import sqlite3

import numpy as np
import torch

aexp = importlib.import_module("deep-image-matching_amd.async_export")
export = importlib.import_module("deep-image-matching_amd.export")


def _fake_batch(rng, B, cap, D=256):
    kp = torch.from_numpy(rng.random((B, cap, 2)).astype(np.float32) * 500)
    sc = torch.from_numpy(rng.random((B, cap)).astype(np.float32))
    de = torch.from_numpy(rng.standard_normal((B, cap, D)).astype(np.float32))
    n = torch.from_numpy(rng.integers(10, cap, B).astype(np.int32))
    return kp, sc, de, n


def test_pack_features_kernel_is_numpy_astype_float16(emu_lib):
    """dim_op_pack_features_f16 == save_features_h5's host conversion (EB:60-67: astype(float16), round-to-nearest-even, overflow
    to inf) + the (N, D) -> (D, N) transpose + un-padding, bit for bit; ragged counts incl. 0 and cap; tile ids."""
    import ctypes
    rng = np.random.default_rng(5)
    B, cap, D = 4, 150, 128
    kp = torch.from_numpy((rng.random((B, cap, 2)) * 3000).astype(np.float32))
    kp[0, 3] = torch.tensor([70000.0, 1e-8])                                       # overflow -> inf, underflow -> 0 / subnormal
    sc = torch.from_numpy(rng.random((B, cap)).astype(np.float32) * 1e-3)
    de = torch.from_numpy(rng.standard_normal((B, cap, D)).astype(np.float32))
    de[1, 7, 5] = 2049.0                                                          # a tie: rounds to even (2048)
    n = torch.tensor([150, 0, 77, 64], dtype=torch.int32)
    ti = torch.from_numpy(rng.integers(0, 16, (B, cap)).astype(np.int32))
    emu_lib.dim_pack_features_slot_halves.restype = ctypes.c_size_t
    slot = emu_lib.dim_pack_features_slot_halves(cap, D)
    assert slot == cap * (4 + D)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    for tile in (ti, None):
        out = torch.full((B * slot,), 123.0, dtype=torch.float16)
        assert emu_lib.dim_op_pack_features_f16(p(kp), p(sc), p(de), p(n), p(tile), B, cap, D, p(out), None) == 0, emu_lib.dim_last_error()
        o = out.numpy().reshape(B, slot)
        for b in range(B):
            k = int(n[b])
            assert np.array_equal(o[b, :2 * k].reshape(k, 2).view(np.uint16), kp[b, :k].numpy().astype(np.float16).view(np.uint16))
            assert np.array_equal(o[b, 2 * cap:2 * cap + k].view(np.uint16), sc[b, :k].numpy().astype(np.float16).view(np.uint16))
            want_t = ti[b, :k].numpy().astype(np.float16) if tile is not None else np.zeros(k, np.float16)
            assert np.array_equal(o[b, 3 * cap:3 * cap + k], want_t)
            assert np.array_equal(o[b, 4 * cap:4 * cap + D * k].reshape(D, k).view(np.uint16), de[b, :k].numpy().T.astype(np.float16).view(np.uint16))
    assert np.isinf(out.numpy().reshape(B, slot)[0, 6]) and emu_lib.dim_op_pack_features_f16(p(kp), p(sc), p(de), p(n), None, B, cap, 100, p(out), None) != 0


def test_async_exporter_equals_synchronous_stores(tmp_path, emu_lib):
    rng = np.random.default_rng(0)
    names = [f"im{i}.jpg" for i in range(6)]
    ex = aexp.AsyncExporter(tmp_path / "async", device="cpu", max_pending=2, lib=emu_lib, feature_workers=3, min_inliers_per_pair=10,
                            min_inlier_ratio_per_pair=0.5, image_names=names)
    sync_f = export.FeatureStore(tmp_path / "sync" / "features.h5")
    for s in range(0, 6, 3):
        kp, sc, de, n = _fake_batch(rng, 3, 64)
        ex.put_features(names[s:s + 3], kp, sc, de, n, [(480, 640)] * 3)
        kp.zero_()      # the exporter must have taken its own copy before returning (the GPU reuses these buffers)
    rng = np.random.default_rng(0)
    for s in range(0, 6, 3):
        kp, sc, de, n = _fake_batch(rng, 3, 64)
        for b in range(3):
            k = int(n[b])
            sync_f.add(names[s + b], {"keypoints": kp[b, :k].numpy(), "descriptors": de[b, :k].numpy().T.copy(), "scores": sc[b, :k].numpy(),
                                       "tile_idx": np.zeros(k, np.float32), "image_size": np.array((480, 640))})
    sync_f.close()
    P, NK = 6, 32
    m = torch.from_numpy(rng.integers(0, 10, (P, NK, 2)).astype(np.int64))
    nm = torch.tensor([20, 5, 0, 32, 30, 24], dtype=torch.int32)
    mask = torch.zeros(P, NK, dtype=torch.uint8)
    mask[0, :20:2] = 1            # 10 of 20: kept (>= 10 inliers, ratio 0.5)
    mask[1, :5] = 1               # < 8 raw matches: skipped (matcher_base.py:287-292)
    mask[3, 3:30] = 1             # 27 of 32: kept
    mask[4, :9] = 1               # 9 inliers < min_inliers_per_pair 10: dropped (matcher_base.py:316-321)
    mask[5, :11] = 1              # 11 of 24 = 0.458 < 0.5: dropped (matcher_base.py:322-327)
    mask[0, 25:] = 1              # beyond n_matches: must be ignored
    pair_names = [(names[0], names[1]), (names[0], names[2]), (names[1], names[2]), (names[2], names[3]), (names[3], names[4]), (names[5], names[4])]
    ex.put_matches(pair_names[:4], m[:4].contiguous(), nm[:4].contiguous(), mask[:4].contiguous())
    ex.put_matches(pair_names[4:], m[4:].contiguous(), nm[4:].contiguous(), mask[4:].contiguous())
    stats = ex.close()
    assert stats["images"] == 6 and stats["pairs"] == 6 and stats["verified_pairs"] == 2 and stats["feature_workers"] == 3
    for nme in names:
        a = export.FeatureStore.read(tmp_path / "async" / "features.h5", nme)
        b = export.FeatureStore.read(tmp_path / "sync" / "features.h5", nme)
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
        assert a["descriptors"].shape[0] == 256 and a["image_size"].tolist() == [480, 640]
    raw = export.MatchStore.read_all(tmp_path / "async" / "raw_matches.h5")
    ver = export.MatchStore.read_all(tmp_path / "async" / "matches.h5")
    assert set(raw) == set(pair_names) and raw[pair_names[0]].shape == (20, 2) and raw[pair_names[2]].shape == (0, 2)
    assert np.array_equal(raw[pair_names[3]], m[3].numpy()) and raw[pair_names[3]].dtype == np.int64
    verify = importlib.import_module("deep-image-matching_amd.verify")
    for p, pn in enumerate(pair_names):     # the device filter == verify.apply_reference_filters == MatcherBase.match's rules
        s = int(nm[p])
        want = verify.apply_reference_filters(m[p, :s].numpy(), mask[p, :s].numpy().astype(bool), 10, 0.5)
        assert (pn in ver) == (want is not None)
        if want is not None:
            assert np.array_equal(ver[pn], want)
    assert set(ver) == {pair_names[0], pair_names[3]}
    db = sqlite3.connect(str(tmp_path / "async" / "database.db"))
    assert db.execute("select count(*) from images").fetchone()[0] == 6
    assert db.execute("select count(*) from matches").fetchone()[0] == 6
    assert db.execute("select count(*) from two_view_geometries").fetchone()[0] == 2
    assert [r[0] for r in db.execute("select name from images order by image_id")] == sorted(names)     # ids as a sorted walk assigns them
    rows, cols, blob = db.execute("select rows, cols, data from keypoints where image_id = 1").fetchone()
    f0 = export.FeatureStore.read(tmp_path / "async" / "features.h5", names[0])
    assert cols == 2 and np.array_equal(np.frombuffer(blob, np.float32).reshape(rows, 2), f0["keypoints"])   # fp16-quantised, io/h5_to_db.py
    # pair (im5, im4): id1 > id2 -> columns swapped in the blob (utils/database.py:263-267)
    pid = export.image_ids_to_pair_id(6, 5)
    r, c, blob = db.execute("select rows, cols, data from matches where pair_id = ?", (pid,)).fetchone()
    assert np.array_equal(np.frombuffer(blob, np.uint32).reshape(r, c), m[5, :24].numpy()[:, ::-1].astype(np.uint32))
    # without the image list the database is written at close from the collected tables: same rows
    ex2 = aexp.AsyncExporter(tmp_path / "late", device="cpu", lib=emu_lib, feature_workers=1, min_inliers_per_pair=10, min_inlier_ratio_per_pair=0.5)
    rng = np.random.default_rng(0)
    for s in range(0, 6, 3):
        kp, sc, de, n = _fake_batch(rng, 3, 64)
        ex2.put_features(names[s:s + 3], kp, sc, de, n, [(480, 640)] * 3)
    ex2.put_matches(pair_names, m, nm, mask)
    ex2.close()
    db2 = sqlite3.connect(str(tmp_path / "late" / "database.db"))
    for table in ("images", "keypoints", "matches", "two_view_geometries"):
        key = "pair_id" if table in ("matches", "two_view_geometries") else "image_id"
        assert db.execute(f"select * from {table} order by {key}").fetchall() == db2.execute(f"select * from {table} order by {key}").fetchall(), table


def test_end_to_end_runner_on_the_emulator(emu_lib, tmp_path):
    sp = importlib.import_module("deep-image-matching_amd.superpoint_hip")
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    verify = importlib.import_module("deep-image-matching_amd.verify")
    pl = importlib.import_module("deep-image-matching_amd.pipeline")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    cfg = {"nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 64, "remove_borders": 2}
    conf = {"n_layers": 2, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
    ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1), cfg, max_batch=2, max_hw=(48, 64), capacity=64, device="cpu", lib=emu_lib)
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, n_layers=2), conf, max_pairs=2, max_kpts=64, device="cpu", lib=emu_lib)
    ver = verify.DeviceVerifier(threshold=3.0, iters=256, device="cpu", lib=emu_lib)
    names = ["a.jpg", "b.jpg", "c.jpg"]
    imgs = torch.rand(3, 48, 64, generator=torch.Generator().manual_seed(0))
    pairs = pl.exhaustive_pairs(3)
    r = aexp.EndToEndRunner(ext, mat, ver, aexp.AsyncExporter(tmp_path, device="cpu", lib=emu_lib, image_names=names, min_inliers_per_pair=0,
                                                              min_inlier_ratio_per_pair=0.0)).run(names, imgs, pairs)
    assert r["images"] == 3 and r["pairs"] == 3 and r["end_to_end_pairs_per_s"] > 0 and r["kernel_path_pairs_per_s"] >= r["end_to_end_pairs_per_s"]
    raw = export.MatchStore.read_all(tmp_path / "raw_matches.h5")
    assert set(raw) == {("a.jpg", "b.jpg"), ("a.jpg", "c.jpg"), ("b.jpg", "c.jpg")}
    assert sum(len(v) for v in raw.values()) == r["raw_matches"]
    f = export.FeatureStore.read(tmp_path / "features.h5", "b.jpg")
    kp, _, de, n = ext.extract_batch(imgs[1:2].contiguous())
    k = int(n[0])
    assert np.array_equal(f["keypoints"], kp[0, :k].numpy().astype(np.float16).astype(np.float32))
    assert r["guard_reruns"] == 0 and np.array_equal(f["descriptors"], de[0, :k].numpy().T.astype(np.float16).astype(np.float32))


def test_h5py_path_precompressed_chunks_read_back_by_the_reference(emu_lib, tmp_path):
    """With h5py present the deflate pool pre-compresses and ONE writer stores the chunks (write_direct_chunk).  Executed against the
    h5py look-alike of tests/refstubs.py (h5py itself is absent from this image) and read back through the REFERENCE's own
    io/h5.py get_features / get_matches; build container only."""
    import pytest
    from tests import refstubs
    if not refstubs.available():
        pytest.skip("/root/reference not present")
    added = refstubs.install()
    try:
        importlib.reload(export)
        assert export.HAVE_H5PY
        rng = np.random.default_rng(3)
        names = ["b.png", "a.png", "c.png"]
        ex = aexp.AsyncExporter(tmp_path, device="cpu", lib=emu_lib, feature_workers=2, image_names=names, min_inliers_per_pair=3, min_inlier_ratio_per_pair=0.1)
        assert ex.features.use_h5 and len(ex._fstores) == 1
        kp, sc, de, n = _fake_batch(rng, 3, 48)
        n[1] = 0                                                     # an image without keypoints: empty datasets
        ex.put_features(names, kp, sc, de, n, [(300, 400)] * 3)
        m = torch.from_numpy(rng.integers(0, 10, (1, 16, 2)).astype(np.int64))
        ex.put_matches([("b.png", "c.png")], m, torch.tensor([12], dtype=torch.int32), torch.ones(1, 16, dtype=torch.uint8))
        ex.close()
        h5 = importlib.import_module("deep_image_matching.io.h5")
        for b, nme in enumerate(names):
            f = h5.get_features(tmp_path / "features.h5", nme)
            k = int(n[b])
            assert f["keypoints"].dtype == np.float32 and np.array_equal(f["keypoints"], kp[b, :k].numpy().astype(np.float16).astype(np.float32))
            assert np.array_equal(f["descriptors"], de[b, :k].numpy().T.astype(np.float16).astype(np.float32)) and f["descriptors"].shape == (256, k)
            assert np.array_equal(f["scores"], sc[b, :k].numpy().astype(np.float16).astype(np.float32)) and f["image_size"].tolist() == [300, 400]
        got = h5.get_matches(tmp_path / "matches.h5", "b.png", "c.png")
        assert np.array_equal(got, m[0, :12].numpy())
    finally:
        refstubs.uninstall(added)
        importlib.reload(export)
        assert not export.HAVE_H5PY


def test_writer_failure_surfaces_instead_of_hanging_the_producer(tmp_path, emu_lib):
    """a writer thread that raises (disk full ...) must not leak its pinned ring buffer — with max_pending
    buffers leaked the producer used to block forever in _Ring.acquire.  Now the buffers come back in a finally block, acquire
    polls the error flag, and close() closes the containers before it reports the first error."""
    import threading
    rng = np.random.default_rng(1)
    names = [f"im{i}.jpg" for i in range(8)]
    ex = aexp.AsyncExporter(tmp_path / "fail", device="cpu", max_pending=1, lib=emu_lib, feature_workers=2, image_names=names)
    boom = OSError(28, "No space left on device")

    def failing(*a, **k):
        raise boom

    ex._write_features = failing
    ex._write_matches = failing
    outcome = {}

    def producer():
        try:
            for rep in range(6):       # far more batches than ring buffers: every one needs a buffer a failed writer must have returned
                kp, sc, de, n = _fake_batch(rng, 2, 16)
                ex.put_features(names[:2], kp, sc, de, n, [(48, 64)] * 2)
                m = torch.zeros(2, 8, 2, dtype=torch.int64)
                ex.put_matches([(names[0], names[1]), (names[1], names[2])], m, torch.tensor([3, 4], dtype=torch.int32))
            outcome["producer"] = "finished"
        except RuntimeError as e:
            outcome["producer"] = e
        try:
            ex.close()
            outcome["close"] = "no error"
        except RuntimeError as e:
            outcome["close"] = e

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    th.join(timeout=60)
    assert not th.is_alive(), "the producer hangs behind a failed writer"
    assert isinstance(outcome["producer"], RuntimeError) and outcome["producer"].__cause__ is boom     # surfaced by _check, not swallowed
    assert isinstance(outcome["close"], RuntimeError) and outcome["close"].__cause__ is boom
