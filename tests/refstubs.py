"""TEST INFRASTRUCTURE: import the REFERENCE's own base classes in a container that lacks their third-party imports.

``deep_image_matching`` cannot be imported here (cv2 / h5py / rasterio / kornia / exifread / pycolmap / matplotlib are
absent and its package ``__init__`` files import every extractor and matcher).  The files the MI355X plugins actually
plug into — ``extractors/extractor_base.py``, ``matchers/matcher_base.py``, ``config.py``, ``constants.py``, ``io/h5.py``,
``utils/{tiling,image,geometric_verification,timer,logger}.py`` — are plain Python, so this module

  1. seeds ``sys.modules`` with small stand-ins for the absent third-party packages (an on-disk, pickle-backed ``h5py``
     look-alike; ``rasterio.open`` / ``cv2.cvtColor`` on PIL / numpy; ``kornia.contrib.compute_padding`` restated from the
     reference's own tests/test_tiling.py known answers; ``cv2.findFundamentalMat`` delegating to a callable the test
     supplies), and
  2. registers ``deep_image_matching`` and its sub-packages as EMPTY package objects whose ``__path__`` points into
     ``/root/reference/src`` — so every ``from ..config import Config`` in the reference files resolves to the reference's
     own file, unmodified, while the heavyweight ``__init__`` files are never executed.

Nothing is copied: the reference sources are executed from where they lie.  Only tests may import this module, and
only in the build container (``available()`` is False on the GPU box, where /root/reference does not exist).
"""
from __future__ import annotations

import importlib
import pickle
import sys
import types
from pathlib import Path

import numpy as np

REF_SRC = Path("/root/reference/src")
PKG = "deep_image_matching"


def available() -> bool:
    return (REF_SRC / PKG / "extractors" / "extractor_base.py").exists()


# ---------------------------------------------------------------------------------------------------------------
# h5py look-alike: groups / datasets in nested dicts, persisted with pickle at close.  Covers exactly what
# extractor_base.save_features_h5, io/h5.get_features / get_matches and matcher_base.match use.
class _Dataset:
    def __init__(self, parent, name, data):
        self.parent, self.name, self._a = parent, name, data

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._a, dtype=dtype)

    def __getitem__(self, key):
        if isinstance(self._a, (bytes, str)):
            return self._a if isinstance(self._a, bytes) else self._a.encode()
        return np.asarray(self._a)[key]

    @property
    def shape(self):
        return np.asarray(self._a).shape

    @property
    def dtype(self):
        return np.asarray(self._a).dtype


class _DatasetID:
    """dset.id.write_direct_chunk(offsets, raw): the chunk arrives already filtered — for a gzip dataset that is a zlib stream,
    which HDF5 would store as is; the look-alike inflates it to prove the bytes are a valid stream of the right size."""

    def __init__(self, store, key, compression, chunks):
        self._s, self._k, self._c, self._chunks = store, key, compression, chunks

    def write_direct_chunk(self, offsets, raw, filter_mask=0):
        import zlib
        a = self._s[self._k]
        assert self._c == "gzip" and tuple(self._chunks) == a.shape and all(o == 0 for o in offsets), "single-chunk gzip datasets only"
        buf = zlib.decompress(raw)
        assert len(buf) == a.nbytes, (len(buf), a.nbytes)
        self._s[self._k] = np.frombuffer(buf, dtype=a.dtype).reshape(a.shape).copy()


class _Group:
    def __init__(self, store: dict, name: str = "/", parent=None):
        self._s, self.name, self.parent = store, name, parent

    def _wrap(self, k, v):
        path = (self.name.rstrip("/") + "/" + k)
        return _Group(v, path, self) if isinstance(v, dict) else _Dataset(self, path, v)

    def __contains__(self, k):
        return k in self._s

    def __getitem__(self, k):
        return self._wrap(k, self._s[k])

    def __delitem__(self, k):
        del self._s[k]

    def keys(self):
        return self._s.keys()

    def items(self):
        return [(k, self._wrap(k, v)) for k, v in self._s.items()]

    def __iter__(self):
        return iter(self._s)

    def __len__(self):
        return len(self._s)

    def create_group(self, k):
        if k in self._s:
            raise ValueError(f"group {k} exists")
        self._s[k] = {}
        return self[k]

    def require_group(self, k):
        self._s.setdefault(k, {})
        return self[k]

    def create_dataset(self, k, data=None, dtype=None, compression=None, compression_opts=None, shape=None, chunks=None, **_):
        if k in self._s:
            raise ValueError(f"Unable to create dataset (name already exists): {k}")
        if isinstance(data, (str, bytes)):
            self._s[k] = data
        elif data is None:   # h5py: an empty dataset of the given shape, filled later (here: by write_direct_chunk)
            self._s[k] = np.zeros(shape, dtype=dtype)
            ds = self[k]
            ds.id = _DatasetID(self._s, k, compression, chunks)
            return ds
        else:
            self._s[k] = np.array(data, dtype=dtype) if dtype is not None else np.array(data)
        return self[k]

    def visititems(self, fn):
        for k, v in self.items():
            fn(k, v)
            if isinstance(v, _Group):
                v.visititems(fn)


class _File(_Group):
    def __init__(self, path, mode="r", libver=None, **_):
        self._path, self._mode = Path(path), mode
        store = {}
        if self._path.exists() and mode != "w":
            store = pickle.loads(self._path.read_bytes())
        elif mode == "r":
            raise OSError(f"Unable to open file (file {path} does not exist)")
        super().__init__(store, "/")

    def close(self):
        if self._mode != "r":
            self._path.write_bytes(pickle.dumps(self._s))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _h5py_module():
    m = types.ModuleType("h5py")
    m.File, m.Group, m.Dataset = _File, _Group, _Dataset
    return m


# ---------------------------------------------------------------------------------------------------------------
def _rasterio_module():
    from PIL import Image

    class _Src:
        def __init__(self, path):
            self._im = Image.open(path)

        def read(self, band=None):
            a = np.asarray(self._im)
            a = a[None] if a.ndim == 2 else np.transpose(a, (2, 0, 1))   # (bands, rows, cols)
            return a if band is None else a[band - 1]

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    m = types.ModuleType("rasterio")
    m.open = lambda path, *a, **k: _Src(path)
    return m


def _cv2_module(find_fundamental=None):
    m = types.ModuleType("cv2")
    for i, n in enumerate(["INTER_NEAREST", "INTER_LINEAR", "INTER_CUBIC", "INTER_AREA", "LMEDS", "RANSAC", "RHO", "USAC_DEFAULT",
                           "USAC_PARALLEL", "USAC_FM_8PTS", "USAC_FAST", "USAC_ACCURATE", "USAC_PROSAC", "USAC_MAGSAC",
                           "COLOR_BGR2GRAY", "COLOR_RGB2GRAY", "IMREAD_GRAYSCALE", "IMREAD_COLOR", "COLOR_GRAY2BGR", "COLOR_BGR2RGB",
                           "FONT_HERSHEY_SIMPLEX", "LINE_AA", "IMWRITE_JPEG_QUALITY"]):
        setattr(m, n, 100 + i)

    def cvtColor(img, code):
        if code in (m.COLOR_BGR2GRAY, m.COLOR_RGB2GRAY):  # OpenCV's fixed-point BGR2GRAY for 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14
            a = img.astype(np.int64)
            c0, c2 = (1868, 4899) if code == m.COLOR_BGR2GRAY else (4899, 1868)
            return ((a[..., 0] * c0 + a[..., 1] * 9617 + a[..., 2] * c2 + 8192) >> 14).astype(img.dtype)
        raise NotImplementedError(code)

    def findFundamentalMat(p0, p1, method=None, ransacReprojThreshold=3.0, confidence=0.99, maxIters=10000):
        if find_fundamental is None:
            raise RuntimeError("cv2.findFundamentalMat is not available in this container")
        return find_fundamental(p0, p1, method, ransacReprojThreshold, confidence, maxIters)

    m.cvtColor, m.findFundamentalMat = cvtColor, findFundamentalMat
    return m


def _kornia_module():
    k, contrib = types.ModuleType("kornia"), types.ModuleType("kornia.contrib")
    k.__version__ = "0.8.1"

    def compute_padding(original_size, window_size, stride=None):
        """kornia 0.8.1 contrib.compute_padding (symmetric padding up to the next multiple of the window), pinned by the
        reference's tests/test_tiling.py:57-88 (100 px, window 40 -> (10, 10, 10, 10))."""
        out = []
        for o, w in zip(original_size, window_size):
            pad = (-o) % w
            out += [pad // 2, pad - pad // 2]   # (top, bottom) / (left, right): the odd pixel goes to the far side
        return tuple(out)

    contrib.compute_padding = compute_padding
    k.contrib = contrib
    return k, contrib


class _Dummy:
    """Stand-in for a download-only model (the base class's own tile preselection networks)."""

    def __init__(self, *a, **k):
        pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


def install(find_fundamental=None):
    """Make ``deep_image_matching.extractors.extractor_base`` / ``.matchers.matcher_base`` importable.  Returns the names
    added to sys.modules (pass them to ``uninstall``)."""
    assert available(), "/root/reference is not present"
    added = []

    def put(name, mod):
        if name not in sys.modules:
            added.append(name)
        sys.modules[name] = mod

    put("h5py", _h5py_module())
    put("rasterio", _rasterio_module())
    put("cv2", _cv2_module(find_fundamental))
    k, kc = _kornia_module()
    put("kornia", k)
    put("kornia.contrib", kc)
    put("exifread", types.ModuleType("exifread"))
    root = REF_SRC / PKG
    for sub in ("", ".utils", ".io", ".extractors", ".matchers", ".thirdparty", ".thirdparty.hloc", ".thirdparty.hloc.extractors",
                ".thirdparty.LightGlue"):
        mod = types.ModuleType(PKG + sub)
        mod.__path__ = [str(root / sub.strip(".").replace(".", "/"))] if sub else [str(root)]
        mod.__package__ = PKG + sub
        put(PKG + sub, mod)
    # constants.py does `from .utils import Timer, setup_logger`: provide them from the reference's own files
    utils = sys.modules[PKG + ".utils"]
    utils.Timer = importlib.import_module(PKG + ".utils.timer").Timer
    utils.setup_logger = importlib.import_module(PKG + ".utils.logger").setup_logger
    added += [PKG + ".utils.timer", PKG + ".utils.logger"]
    # matcher_base.py imports the base class's preselection networks (URL downloads) and the matplotlib visualisers
    hs = types.ModuleType(PKG + ".thirdparty.hloc.extractors.superpoint")
    hs.SuperPoint = _Dummy
    put(hs.__name__, hs)
    lgp = types.ModuleType(PKG + ".thirdparty.LightGlue.lightglue")
    lgp.LightGlue = _Dummy
    put(lgp.__name__, lgp)
    viz = types.ModuleType(PKG + ".visualization")
    viz.viz_matches_cv2 = viz.viz_matches_mpl = lambda *a, **k: None
    put(viz.__name__, viz)
    return added


def uninstall(added):
    for name in list(sys.modules):
        if name in added or name == PKG or name.startswith(PKG + "."):
            sys.modules.pop(name, None)
    for name in ("h5py", "rasterio", "cv2", "kornia", "kornia.contrib", "exifread"):
        if name in added:
            sys.modules.pop(name, None)
