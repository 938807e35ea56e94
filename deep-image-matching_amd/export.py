"""Output formats behind the boundary: features / matches containers and the COLMAP database.

Restatement of the reference's writers so that results produced by the batched MI355X pipeline
(``pipeline.PairMatchingPipeline``) land in exactly the artefacts the rest of deep-image-matching
and COLMAP consume:

* ``features.h5``   group <image name> -> keypoints (N,2), descriptors (D,N), scores (N,),
                    tile_idx (N,), image_size (2,), ALL float16, gzip-9, libver="latest"
                    (extractors/extractor_base.py:56-99, io/h5.py:45-89; quirk Q6: keypoints are
                    quantised to fp16 too).
* ``raw_matches.h5`` / ``matches.h5``  group <img0> -> dataset <img1> int64 (S,2)
                    (matchers/matcher_base.py:282-285,337-339).
* ``database.db``   COLMAP schema (utils/database.py:40-110), keypoints float32 blob, raw matches
                    into ``matches`` (uint32, swapped when id1 > id2), verified matches into
                    ``two_view_geometries`` with config 2 and identity F/E/H (utils/database.py:242-311,
                    io/h5_to_db.py:264-340), pair_id = id1*(2^31-1)+id2 (:113-116).

h5py is an optional dependency: when it is missing (as in the build container) the containers
are written as ``.npz`` mirrors with the same group/dataset names, dtypes and shapes
("<group>/<dataset>" keys) and ``npz_to_h5`` converts them where h5py exists.  The COLMAP database
needs only the standard library.
"""
from __future__ import annotations

import sqlite3
from pathlib import Path
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

try:
    import h5py  # noqa: F401

    HAVE_H5PY = True
except Exception:  # noqa: BLE001
    HAVE_H5PY = False

MAX_IMAGE_ID = 2**31 - 1  # utils/database.py:38

_SCHEMA = [  # utils/database.py:40-99
    """CREATE TABLE IF NOT EXISTS cameras (camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, model INTEGER NOT NULL,
       width INTEGER NOT NULL, height INTEGER NOT NULL, params BLOB, prior_focal_length INTEGER NOT NULL)""",
    f"""CREATE TABLE IF NOT EXISTS images (image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, name TEXT NOT NULL UNIQUE,
       camera_id INTEGER NOT NULL, prior_qw REAL, prior_qx REAL, prior_qy REAL, prior_qz REAL, prior_tx REAL, prior_ty REAL,
       prior_tz REAL, CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < {MAX_IMAGE_ID}),
       FOREIGN KEY(camera_id) REFERENCES cameras(camera_id))""",
    """CREATE TABLE IF NOT EXISTS keypoints (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,
       data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)""",
    """CREATE TABLE IF NOT EXISTS descriptors (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,
       data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE)""",
    "CREATE TABLE IF NOT EXISTS matches (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL, data BLOB)",
    """CREATE TABLE IF NOT EXISTS two_view_geometries (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
       cols INTEGER NOT NULL, data BLOB, config INTEGER NOT NULL, F BLOB, E BLOB, H BLOB, qvec BLOB, tvec BLOB)""",
    "CREATE UNIQUE INDEX IF NOT EXISTS index_name ON images(name)",
]
_CAMERA_MODELS = {"simple-pinhole": 0, "pinhole": 1, "simple-radial": 2, "opencv": 4}  # io/h5_to_db.py:124-146


def image_ids_to_pair_id(id1: int, id2: int) -> int:
    if id1 > id2:
        id1, id2 = id2, id1
    return id1 * MAX_IMAGE_ID + id2


# ---------------------------------------------------------------------------------------------
# features / matches containers
# ---------------------------------------------------------------------------------------------
def features_to_half(features: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """save_features_h5(as_half=True) (EB:60-67): every float32 ndarray -> float16; everything is
    then stored with dtype float16 (EB:80-86), image_size included."""
    return {k: np.asarray(v).astype(np.float16) for k, v in features.items() if isinstance(v, np.ndarray)}


class _NpzAppender:
    """Incremental writer of an .npz (a zip of .npy members): every ``add`` deflates and appends one member, so nothing
    is held back for ``close`` (np.savez_compressed would write everything at the end — on the critical path of a run
    that is otherwise overlapped with the GPU).  np.load reads the result like any other .npz."""

    def __init__(self, path: Path, compresslevel: Optional[int] = 9):
        import zipfile

        self.path = Path(path)
        self.path.parent.mkdir(parents=True, exist_ok=True)
        # compresslevel 9 = the gzip-9 of features.h5 (EB:80-86); None = stored, like the uncompressed match datasets
        # h5py writes at MB:282-285,337-339
        if compresslevel is None:
            self._zf = zipfile.ZipFile(str(self.path), "w", zipfile.ZIP_STORED, allowZip64=True)
        else:
            self._zf = zipfile.ZipFile(str(self.path), "w", zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=compresslevel)
        self.keys = set()

    def add(self, key: str, arr: np.ndarray):
        with self._zf.open(key + ".npy", "w", force_zip64=True) as f:
            np.lib.format.write_array(f, np.asanyarray(arr), allow_pickle=False)
        self.keys.add(key)

    def close(self):
        self._zf.close()


def _npz_shards(path: Path):
    """features.npz plus the shard files parallel writers add next to it (features.shard1.npz, ...)."""
    base = Path(path).with_suffix(".npz")
    return [p for p in [base] + sorted(base.parent.glob(base.stem + ".shard*.npz")) if p.exists()]


class FeatureStore:
    """features.h5 writer/reader (h5py) or its .npz mirror.  ``shard`` > 0 writes ``<stem>.shard<k>.npz`` next to the main
    mirror so that several writer threads can compress in parallel (zlib releases the GIL); readers see the union."""

    def __init__(self, path: Path, shard: int = 0):
        self.path = Path(path)
        self.use_h5 = HAVE_H5PY and self.path.suffix == ".h5"
        self._npz = None
        if not self.use_h5:
            base = self.path.with_suffix(".npz")
            self._npz = _NpzAppender(base if shard == 0 else base.with_name(f"{base.stem}.shard{shard}.npz"))

    def add(self, im_name: str, features: Dict[str, np.ndarray]):
        self.add_half(im_name, features_to_half(features))

    def add_precompressed(self, im_name: str, half: Dict[str, np.ndarray], blobs: Dict[str, bytes]):
        """h5py only: datasets whose single chunk was deflated by the caller (zlib stream = HDF5's gzip filter format) are
        stored with write_direct_chunk, so the time under h5py's global lock is a byte copy.  Readers see ordinary gzip-9
        float16 datasets (one chunk per dataset instead of h5py's guessed chunk shape)."""
        import h5py

        with h5py.File(str(self.path), "a", libver="latest") as fd:
            if im_name in fd:
                del fd[im_name]
            grp = fd.create_group(im_name)
            for k, v in half.items():
                if v.size == 0:
                    grp.create_dataset(k, data=v, dtype=np.float16)
                    continue
                ds = grp.create_dataset(k, shape=v.shape, dtype=np.float16, chunks=v.shape, compression="gzip", compression_opts=9)
                ds.id.write_direct_chunk((0,) * v.ndim, blobs[k])

    def add_half(self, im_name: str, half: Dict[str, np.ndarray]):
        """``half``: the float16 arrays exactly as they are stored (save_features_h5 after its as_half conversion)."""
        if self.use_h5:
            import h5py

            with h5py.File(str(self.path), "a", libver="latest") as fd:
                if im_name in fd:
                    del fd[im_name]
                grp = fd.create_group(im_name)
                for k, v in half.items():
                    grp.create_dataset(k, data=v, dtype=np.float16, compression="gzip", compression_opts=9)
        else:
            for k, v in half.items():
                self._npz.add(f"{im_name}/{k}", v)

    def close(self):
        if self._npz is not None:
            self._npz.close()

    @staticmethod
    def read(path: Path, im_name: str) -> Dict[str, np.ndarray]:
        """io/h5.py:45-89 get_features: keypoints/descriptors/scores/tile_idx -> float32, image_size -> int32."""
        path = Path(path)
        if path.suffix == ".h5" and HAVE_H5PY:
            import h5py

            with h5py.File(str(path), "r") as fd:
                raw = {k: np.array(v) for k, v in fd[im_name].items()}
        else:
            raw = {}
            for shard in _npz_shards(path):
                z = np.load(str(shard))
                raw.update({k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(im_name + "/")})
            if not raw:
                raise KeyError(f"Cannot find image {im_name} in {path}")
        out = {k: raw[k].astype(np.float32) for k in ("keypoints", "descriptors", "scores", "tile_idx") if k in raw}
        if "image_size" in raw:
            out["image_size"] = raw["image_size"].astype(np.int32)
        return out


class MatchStore:
    """raw_matches.h5 / matches.h5 writer (group img0 -> dataset img1, int64 (S,2)) or .npz mirror."""

    def __init__(self, path: Path):
        self.path = Path(path)
        self.use_h5 = HAVE_H5PY and self.path.suffix == ".h5"
        self._npz = None if self.use_h5 else _NpzAppender(self.path.with_suffix(".npz"), compresslevel=None)

    def add(self, img0: str, img1: str, matches: np.ndarray):
        m = np.asarray(matches).reshape(-1, 2).astype(np.int64)
        if self.use_h5:
            import h5py

            with h5py.File(str(self.path), "a", libver="latest") as fd:
                grp = fd.require_group(img0)
                grp.create_dataset(img1, data=m)  # raises if the pair exists, like MB:282-285
        else:
            key = f"{img0}/{img1}"
            if key in self._npz.keys:
                raise ValueError(f"Unable to create dataset (name already exists): {key}")
            self._npz.add(key, m)

    def close(self):
        if self._npz is not None:
            self._npz.close()

    @staticmethod
    def read_all(path: Path) -> Dict[Tuple[str, str], np.ndarray]:
        path = Path(path)
        if path.suffix == ".h5" and HAVE_H5PY:
            import h5py

            out = {}
            with h5py.File(str(path), "r") as fd:
                for a in fd:
                    for b in fd[a]:
                        out[(a, b)] = fd[a][b][()]
            return out
        z = np.load(str(path.with_suffix(".npz")))
        return {tuple(k.split("/", 1)): z[k] for k in z.files}


def npz_to_h5(npz_path: Path, h5_path: Path, half: bool) -> None:
    """Convert a mirror written without h5py into the real container (run where h5py exists)."""
    import h5py

    z = np.load(str(npz_path))
    with h5py.File(str(h5_path), "a", libver="latest") as fd:
        for key in z.files:
            g, d = key.split("/", 1)
            grp = fd.require_group(g)
            if half:
                grp.create_dataset(d, data=z[key], dtype=np.float16, compression="gzip", compression_opts=9)
            else:
                grp.create_dataset(d, data=z[key])


# ---------------------------------------------------------------------------------------------
# COLMAP database
# ---------------------------------------------------------------------------------------------
class ColmapDatabase:
    """utils/database.py:133-311 on the standard library's sqlite3."""

    def __init__(self, path: Path, overwrite: bool = True):
        path = Path(path)
        if overwrite and path.exists():  # io/h5_to_db.py:80-85
            path.unlink()
        self.db = sqlite3.connect(str(path))
        for stmt in _SCHEMA:
            self.db.execute(stmt)
        self.db.commit()

    def add_camera(self, model: str, width: int, height: int, params, prior_focal_length: bool = False, camera_id: Optional[int] = None) -> int:
        if model not in _CAMERA_MODELS:
            raise RuntimeError(f"Invalid camera model {model}")
        cur = self.db.execute("INSERT INTO cameras VALUES (?, ?, ?, ?, ?, ?)",
                              (camera_id, str(_CAMERA_MODELS[model]), int(width), int(height), np.asarray(params, np.float64).tobytes(),
                               bool(prior_focal_length)))
        return cur.lastrowid

    def add_default_camera(self, model: str, width: int, height: int, focal_35mm: Optional[float] = None, camera_id: Optional[int] = None) -> int:
        """io/h5_to_db.py:116-149,342-386: focal from EXIF FocalLengthIn35mmFilm if known, else the
        1.2 * max(w, h) prior; principal point at the image centre."""
        focal = (focal_35mm / 35.0 if focal_35mm else 1.2) * max(width, height)
        params = {"simple-pinhole": [focal, width / 2, height / 2], "pinhole": [focal, focal, width / 2, height / 2],
                  "simple-radial": [focal, width / 2, height / 2, 0.1],
                  "opencv": [focal, focal, width / 2, height / 2, 0.0, 0.0, 0.0, 0.0]}[model]
        return self.add_camera(model, width, height, params, camera_id=camera_id)

    def add_image(self, name: str, camera_id: int, image_id: Optional[int] = None) -> int:
        """``image_id`` / ``camera_id`` None = AUTOINCREMENT (utils/database.py:196-225 passes None too); explicit ids let rows be
        inserted out of order while a run is in flight and still land on the ids a sorted walk would assign."""
        cur = self.db.execute("INSERT INTO images VALUES (?, ?, ?, ?, ?, ?, ?, ?, ?, ?)", (image_id, name, camera_id, 0, 0, 0, 0, 0, 0, 0))
        return cur.lastrowid

    def add_keypoints(self, image_id: int, keypoints: np.ndarray):
        assert keypoints.ndim == 2 and keypoints.shape[1] in (2, 4, 6)
        k = np.asarray(keypoints, np.float32)
        self.db.execute("INSERT INTO keypoints VALUES (?, ?, ?, ?)", (image_id,) + k.shape + (k.tobytes(),))

    def add_matches(self, id1: int, id2: int, matches: np.ndarray):
        assert matches.ndim == 2 and matches.shape[1] == 2
        if id1 > id2:
            matches = matches[:, ::-1]
        m = np.asarray(matches, np.uint32)
        self.db.execute("INSERT INTO matches VALUES (?, ?, ?, ?)", (image_ids_to_pair_id(id1, id2),) + m.shape + (m.tobytes(),))

    def add_two_view_geometry(self, id1: int, id2: int, matches: np.ndarray, config: int = 2):
        assert matches.ndim == 2 and matches.shape[1] == 2
        if id1 > id2:
            matches = matches[:, ::-1]
        m = np.asarray(matches, np.uint32)
        eye = np.eye(3, dtype=np.float64).tobytes()
        self.db.execute("INSERT INTO two_view_geometries VALUES (?, ?, ?, ?, ?, ?, ?, ?, ?, ?)",
                        (image_ids_to_pair_id(id1, id2),) + m.shape + (m.tobytes(), config, eye, eye, eye,
                                                                       np.array([1.0, 0.0, 0.0, 0.0]).tobytes(), np.zeros(3).tobytes()))

    def commit(self):
        self.db.commit()

    def close(self):
        self.db.commit()
        self.db.close()


def export_to_colmap(database_path: Path, image_names: Iterable[str], image_wh: Dict[str, Tuple[int, int]],
                     keypoints: Dict[str, np.ndarray], raw_matches: Dict[Tuple[str, str], np.ndarray],
                     verified_matches: Optional[Dict[Tuple[str, str], np.ndarray]] = None, camera_model: str = "simple-radial",
                     single_camera: bool = False) -> Dict[str, int]:
    """io/h5_to_db.py:44-113 from in-memory tables (what FeatureStore/MatchStore hold): cameras +
    images + keypoints, raw matches -> ``matches``, verified matches -> ``two_view_geometries``.
    Duplicate pairs are skipped like io/h5_to_db.py:286-292.  Returns {image name: image_id}."""
    db = ColmapDatabase(database_path)
    ids: Dict[str, int] = {}
    cam0 = None
    for name in image_names:
        w, h = image_wh[name]
        if single_camera:
            cam0 = cam0 if cam0 is not None else db.add_default_camera(camera_model, w, h)
            cam = cam0
        else:
            cam = db.add_default_camera(camera_model, w, h)
        ids[name] = db.add_image(name, cam)
        k = keypoints[name]
        if k.ndim >= 2:
            db.add_keypoints(ids[name], k)
    for table, add in ((raw_matches, db.add_matches), (verified_matches or {}, db.add_two_view_geometry)):
        seen = set()
        for (a, b), m in table.items():
            pid = image_ids_to_pair_id(ids[a], ids[b])
            if pid in seen:
                continue
            add(ids[a], ids[b], np.asarray(m).reshape(-1, 2))
            seen.add(pid)
    db.close()
    return ids
