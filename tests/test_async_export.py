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


def test_async_exporter_equals_synchronous_stores(tmp_path):
    rng = np.random.default_rng(0)
    names = [f"im{i}.jpg" for i in range(6)]
    ex = aexp.AsyncExporter(tmp_path / "async", device="cpu", max_pending=2)
    sync_f = export.FeatureStore(tmp_path / "sync" / "features.h5")
    batches = []
    for s in range(0, 6, 3):
        kp, sc, de, n = _fake_batch(rng, 3, 64)
        ex.put_features(names[s:s + 3], kp, sc, de, n, [(480, 640)] * 3)
        kp.zero_()      # the exporter must have taken its own copy before returning (the GPU reuses these buffers)
        batches.append((s, n))
    rng = np.random.default_rng(0)
    for s in range(0, 6, 3):
        kp, sc, de, n = _fake_batch(rng, 3, 64)
        for b in range(3):
            k = int(n[b])
            sync_f.add(names[s + b], {"keypoints": kp[b, :k].numpy(), "descriptors": de[b, :k].numpy().T.copy(), "scores": sc[b, :k].numpy(),
                                       "tile_idx": np.zeros(k, np.float32), "image_size": np.array((480, 640))})
    sync_f.close()
    P, NK = 4, 32
    m = torch.from_numpy(rng.integers(0, 10, (P, NK, 2)).astype(np.int64))
    nm = torch.tensor([20, 5, 0, 32], dtype=torch.int32)
    mask = torch.from_numpy((rng.random((P, NK)) > 0.4).astype(np.uint8))
    pair_names = [(names[0], names[1]), (names[0], names[2]), (names[1], names[2]), (names[2], names[3])]
    ex.put_matches(pair_names, m, nm, mask)
    stats = ex.close()
    assert stats["images"] == 6 and stats["pairs"] == 4
    for nme in names:
        a = export.FeatureStore.read(tmp_path / "async" / "features.h5", nme)
        b = export.FeatureStore.read(tmp_path / "sync" / "features.h5", nme)
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
        assert a["descriptors"].shape[0] == 256 and a["image_size"].tolist() == [480, 640]
    raw = export.MatchStore.read_all(tmp_path / "async" / "raw_matches.h5")
    ver = export.MatchStore.read_all(tmp_path / "async" / "matches.h5")
    assert set(raw) == set(pair_names) and raw[pair_names[0]].shape == (20, 2) and raw[pair_names[2]].shape == (0, 2)
    assert np.array_equal(raw[pair_names[3]], m[3].numpy())
    assert set(ver) == {pair_names[0], pair_names[3]}          # pairs with < 8 raw matches are not verified (matcher_base.py:287-292)
    assert np.array_equal(ver[pair_names[0]], m[0, :20].numpy()[mask[0, :20].numpy().astype(bool)])
    db = sqlite3.connect(str(tmp_path / "async" / "database.db"))
    assert db.execute("select count(*) from images").fetchone()[0] == 6
    assert db.execute("select count(*) from matches").fetchone()[0] == 4
    assert db.execute("select count(*) from two_view_geometries").fetchone()[0] == 2
    rows, cols, blob = db.execute("select rows, cols, data from keypoints where image_id = 1").fetchone()
    assert cols == 2 and len(blob) == rows * 2 * 4


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
    r = aexp.EndToEndRunner(ext, mat, ver, aexp.AsyncExporter(tmp_path, device="cpu")).run(names, imgs, pairs)
    assert r["images"] == 3 and r["pairs"] == 3 and r["end_to_end_pairs_per_s"] > 0 and r["kernel_path_pairs_per_s"] >= r["end_to_end_pairs_per_s"]
    raw = export.MatchStore.read_all(tmp_path / "raw_matches.h5")
    assert set(raw) == {("a.jpg", "b.jpg"), ("a.jpg", "c.jpg"), ("b.jpg", "c.jpg")}
    assert sum(len(v) for v in raw.values()) == r["raw_matches"]
    f = export.FeatureStore.read(tmp_path / "features.h5", "b.jpg")
    kp, _, de, n = ext.extract_batch(imgs[1:2].contiguous())
    k = int(n[0])
    assert np.array_equal(f["keypoints"], kp[0, :k].numpy().astype(np.float16).astype(np.float32))
