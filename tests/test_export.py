"""CPU: output formats (features/matches containers, COLMAP database) follow the reference's layouts."""
import importlib
import sqlite3

import numpy as np

export = importlib.import_module("deep-image-matching_amd.export")


def _feats(n, d, seed):
    r = np.random.RandomState(seed)
    return {"keypoints": (r.rand(n, 2) * 3000).astype(np.float32), "descriptors": r.randn(d, n).astype(np.float32),
            "scores": r.rand(n).astype(np.float32), "tile_idx": np.zeros(n, np.float32), "image_size": np.array([2000, 3000])}


def test_feature_store_roundtrip_is_fp16_like_the_reference(tmp_path):
    st = export.FeatureStore(tmp_path / "features.h5")
    f = _feats(50, 256, 0)
    st.add("img_a.jpg", f)
    st.add("img_b.jpg", _feats(7, 256, 1))
    st.close()
    g = export.FeatureStore.read(tmp_path / "features.h5", "img_a.jpg")
    assert g["keypoints"].dtype == np.float32 and g["keypoints"].shape == (50, 2)
    assert g["descriptors"].shape == (256, 50) and g["image_size"].dtype == np.int32
    # Q6: everything went through float16 (EB:60-67): keypoints above 2048 are quantised
    assert np.array_equal(g["keypoints"], f["keypoints"].astype(np.float16).astype(np.float32))
    assert not np.array_equal(g["keypoints"], f["keypoints"])
    assert g["image_size"].tolist() == [2000, 3000]


def test_match_store_layout_and_duplicate_error(tmp_path):
    st = export.MatchStore(tmp_path / "matches.h5")
    m = np.array([[0, 5], [3, 1]], dtype=np.int32)
    st.add("a.jpg", "b.jpg", m)
    try:
        st.add("a.jpg", "b.jpg", m)
        raise AssertionError("duplicate pair must raise like h5py create_dataset")
    except (ValueError, RuntimeError):
        pass
    st.add("a.jpg", "c.jpg", np.zeros((0, 2)))
    st.close()
    allm = export.MatchStore.read_all(tmp_path / "matches.h5")
    assert allm[("a.jpg", "b.jpg")].dtype == np.int64 and allm[("a.jpg", "b.jpg")].tolist() == [[0, 5], [3, 1]]
    assert allm[("a.jpg", "c.jpg")].shape == (0, 2)


def test_colmap_database_blobs_and_pair_ids(tmp_path):
    names = ["a.jpg", "b.jpg", "c.jpg"]
    wh = {"a.jpg": (640, 480), "b.jpg": (618, 640), "c.jpg": (784, 784)}
    kp = {n: _feats(10 + i, 256, i)["keypoints"] for i, n in enumerate(names)}
    raw = {("a.jpg", "b.jpg"): np.array([[0, 1], [2, 3]]), ("c.jpg", "a.jpg"): np.array([[4, 5]]), ("b.jpg", "a.jpg"): np.array([[9, 9]])}
    ver = {("a.jpg", "b.jpg"): np.array([[0, 1]])}
    ids = export.export_to_colmap(tmp_path / "database.db", names, wh, kp, raw, ver)
    assert ids == {"a.jpg": 1, "b.jpg": 2, "c.jpg": 3}
    db = sqlite3.connect(str(tmp_path / "database.db"))
    tables = {r[0] for r in db.execute("select name from sqlite_master where type='table'")}
    assert {"cameras", "images", "keypoints", "descriptors", "matches", "two_view_geometries"} <= tables
    cam = db.execute("select model,width,height,params,prior_focal_length from cameras where camera_id=1").fetchone()
    p = np.frombuffer(cam[3], np.float64)
    assert int(cam[0]) == 2 and (cam[1], cam[2]) == (640, 480) and np.allclose(p, [1.2 * 640, 320, 240, 0.1]) and cam[4] == 0
    r, c, blob = db.execute("select rows,cols,data from keypoints where image_id=2").fetchone()
    assert (r, c) == (11, 2) and np.array_equal(np.frombuffer(blob, np.float32).reshape(r, c), kp["b.jpg"])
    rows = dict((pid, (r, c, blob)) for pid, r, c, blob in db.execute("select pair_id,rows,cols,data from matches"))
    M = 2**31 - 1
    assert set(rows) == {1 * M + 2, 1 * M + 3}  # (b,a) duplicate of (a,b) skipped; (c,a) stored under (a,c)
    assert np.frombuffer(rows[1 * M + 3][2], np.uint32).reshape(-1, 2).tolist() == [[5, 4]]  # swapped because id1 > id2
    tv = db.execute("select pair_id,rows,cols,data,config,F,qvec from two_view_geometries").fetchall()
    assert len(tv) == 1 and tv[0][0] == 1 * M + 2 and tv[0][4] == 2
    assert np.array_equal(np.frombuffer(tv[0][5], np.float64).reshape(3, 3), np.eye(3)) and np.frombuffer(tv[0][6], np.float64).tolist() == [1, 0, 0, 0]
    assert export.image_ids_to_pair_id(5, 2) == 2 * M + 5
