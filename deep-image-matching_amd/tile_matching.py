"""Tile-wise matching for large-format images, batched on the GPU.

Host-side restatement of ``MatcherBase._match_by_tile`` (matchers/matcher_base.py:362-460) and
``tile_selection`` (matchers/matcher_base.py:989-1342: EXHAUSTIVE / GRID / PRESELECTION / PRESELECTION_AFFINE_TRANSFORM with
the superpoint+lightglue pipeline, every quality level) with two differences that do not change results:

* the reference calls ``_match_pairs`` once per tile pair, sequentially (MB:417-425); here the tiles of
  both images form ONE device feature table and all selected tile pairs go through
  ``LightGlueHIP.match_batch`` in batches (pair -> table-row indirection, no per-pair upload);
* the reference re-runs the preselection SuperPoint on both (re-read, re-decoded) images for every image
  pair (MB:1021-1024, 1072-1074); here the down-sampled features are cached per image, and the whole
  preselection — INTER_AREA down-sampling, SuperPoint (nms 5 / 4000 kpts / thr 0.005), LightGlue
  (depth 0.9 / width 0.95 / filter 0.3), per-tile-pair vote count — stays on the device
  (``dim_op_resize_area_f32``, ``dim_sp_extract``, ``dim_lg_match``, ``dim_op_tile_pair_votes``).

Per-tile geometric verification (``geometric_verification_per_tile``, MB:427-441) is cv2 RANSAC in the
reference and is not rebuilt: with that option set the mixin defers to the base class.
"""
from __future__ import annotations

import ctypes
import logging
from collections import OrderedDict
from itertools import product
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi
from . import weights as _weights
from .lightglue_hip import LightGlueHIP
from .superpoint_hip import SuperPointHIP
from .tiling import compute_padding

logger = logging.getLogger("dim_amd")

# MatcherBase.__init__ (MB:136-149): the preselection networks' settings.  The extractor is hloc's SuperPoint wrapper
# (MB:18), whose defaults add remove_borders 4 and fix_sampling True (thirdparty/hloc/extractors/superpoint.py:24-31, Q3).
PRESELECTION_SP_CONF = {"nms_radius": 5, "max_keypoints": 4000, "keypoint_threshold": 0.005, "remove_borders": 4, "fix_sampling": True}
PRESELECTION_LG_CONF = {"n_layers": 9, "depth_confidence": 0.9, "width_confidence": 0.95, "filter_threshold": 0.3}
_QUALITY_FACTOR = {"HIGHEST": 2, "HIGH": 1, "MEDIUM": 1 / 2, "LOW": 1 / 4, "LOWEST": 1 / 8}  # constants.py:80-86


def get_size_by_quality(quality: str, size: Tuple[int, int]) -> Tuple[int, int]:
    """constants.py:76-88 (pinned against the reference's function by oracle/make_golden.py)."""
    f = _QUALITY_FACTOR[quality]
    return (int(size[0] * f), int(size[1] * f))


def get_features_by_tile(features: dict, tile_idx: int):
    """MB:1380-1391."""
    if "tile_idx" not in features:
        raise KeyError("tile_idx not found in features")
    sel = features["tile_idx"] == tile_idx
    idx = np.where(sel)[0]
    return ({"keypoints": features["keypoints"][sel], "descriptors": features["descriptors"][:, sel],
             "scores": features["scores"][sel], "image_size": features["image_size"]}, idx)


def tile_grid(shape_hw: Tuple[int, int], tile_size, overlap=0) -> Dict[int, Tuple[int, int]]:
    """Tile ids and (x, y) origins of ``Tiler.compute_tiles_by_size`` (utils/tiling.py:62-192) from the
    image shape alone (tile_selection only uses the keys and origins, MB:1036-1041)."""
    win = (tile_size, tile_size) if isinstance(tile_size, int) else (tile_size[1], tile_size[0])  # (H, W)
    ov = (overlap, overlap) if isinstance(overlap, int) else (overlap[1], overlap[0])
    H, W = int(shape_hw[0]), int(shape_hw[1])
    pad = compute_padding((H, W), win)
    stride = (win[0] - ov[0], win[1] - ov[1])
    n_rows = (H + pad[0] + pad[1] - win[0]) // stride[0] + 1
    n_cols = (W + pad[2] + pad[3] - win[1]) // stride[1] + 1
    return {r * n_cols + c: (-pad[2] + c * stride[1], -pad[0] + r * stride[0]) for r in range(n_rows) for c in range(n_cols)}


def select_tile_pairs(method: str, keys0: Sequence[int], keys1: Sequence[int], votes: Optional[np.ndarray] = None,
                      min_matches_per_tile: int = 5) -> List[Tuple[int, int]]:
    """MB:1042-1052 and 1124-1133.  ``votes[a, b]`` is indexed by the sorted tile keys."""
    if method == "EXHAUSTIVE":
        return sorted(product(keys0, keys1))
    if method == "GRID":
        return sorted(zip(keys0, keys1))
    if method == "PRESELECTION":
        if votes is None:
            raise ValueError("PRESELECTION needs the vote table")
        k0, k1 = sorted(keys0), sorted(keys1)
        return sorted((k0[a], k1[b]) for a in range(len(k0)) for b in range(len(k1)) if votes[a, b] > min_matches_per_tile)
    raise ValueError(f"Invalid tile selection method: {method}")   # MB:1335-1336


def _affine_total_least_squares(kp0: np.ndarray, kp1: np.ndarray) -> np.ndarray:
    """The reference's second-stage estimator, skimage.transform.estimate_transform('affine', kp0, kp1) (MB:1441-1445), restated
    from scikit-image's published algorithm (AffineTransform.estimate: Hartley-normalised points, the 6 affine coefficients
    + the homogeneous scale as the right singular vector of the smallest singular value).  skimage is absent here: unpinned."""
    def normalise(p):
        c = p.mean(axis=0)
        rms = np.sqrt(np.mean(np.sum((p - c) ** 2, axis=1)))
        if rms == 0:
            raise ValueError("degenerate points")
        k = np.sqrt(2.0) / rms
        T = np.array([[k, 0, -k * c[0]], [0, k, -k * c[1]], [0, 0, 1.0]])
        return T, (p - c) * k
    T0, a = normalise(np.asarray(kp0, np.float64))
    T1, b = normalise(np.asarray(kp1, np.float64))
    n = len(a)
    A = np.zeros((2 * n, 7))
    A[:n, 0:2], A[:n, 2], A[:n, 6] = a, 1.0, b[:, 0]
    A[n:, 3:5], A[n:, 5], A[n:, 6] = a, 1.0, b[:, 1]
    V = np.linalg.svd(A)[2]
    H = np.eye(3)
    H.flat[:6] = -V[-1, :6] / V[-1, 6]
    H = np.linalg.inv(T1) @ H @ T0
    return (H / H[2, 2])[:2]


def _similarity_ransac(kp0: np.ndarray, kp1: np.ndarray, threshold: float = 4.0, confidence: float = 0.999, max_iters: int = 10000,
                       seed: int = 0) -> Optional[np.ndarray]:
    """Stand-in for cv2.estimateAffinePartial2D(method=RANSAC, ransacReprojThreshold=4, confidence=0.999, maxIters=10000)
    (MB:1436-1440) where OpenCV is absent: the same model class (4-DoF similarity from 2-point samples), threshold and
    adaptive stopping rule, least-squares refit on the inliers.  Deterministic (seeded); NOT result-identical to OpenCV's RNG
    and LM refinement — the transform only feeds rectangle-intersection tests with a >= 2 px margin (MB:1276-1310)."""
    a, b = np.asarray(kp0, np.float64), np.asarray(kp1, np.float64)
    n = len(a)
    rng = np.random.default_rng(seed)

    def fit(p, q):   # q ~ s R p + t, least squares (Umeyama without reflection handling: [a -b; b a] parametrisation)
        pc, qc = p.mean(0), q.mean(0)
        dp, dq = p - pc, q - qc
        den = np.sum(dp * dp)
        if den <= 0:
            return None
        ca = np.sum(dp[:, 0] * dq[:, 0] + dp[:, 1] * dq[:, 1]) / den
        sa = np.sum(dp[:, 0] * dq[:, 1] - dp[:, 1] * dq[:, 0]) / den
        R = np.array([[ca, -sa], [sa, ca]])
        return np.c_[R, qc - R @ pc]

    best, best_mask, iters, it = -1, None, int(max_iters), 0
    thr2 = threshold * threshold
    az, bz = a[:, 0] + 1j * a[:, 1], b[:, 0] + 1j * b[:, 1]       # points as complex numbers: a similarity is z -> r z + t
    while it < iters:
        batch = min(256, iters - it)
        i0 = rng.integers(0, n, batch)
        i1 = (i0 + 1 + rng.integers(0, n - 1, batch)) % n
        # all samples of the batch at once (the per-sample Python loop was the slow part): the 2-point fit is exact,
        # r = (q1 - q0) / (p1 - p0), t = q0 - r p0; residuals of every point under every sample as one (batch, n) array
        dp = az[i1] - az[i0]
        ok = dp != 0
        r = np.where(ok, (bz[i1] - bz[i0]) / np.where(ok, dp, 1.0), 0.0)
        tt = bz[i0] - r * az[i0]
        res = r[:, None] * az[None, :] + tt[:, None] - bz[None, :]
        masks = (res.real ** 2 + res.imag ** 2) <= thr2
        counts = masks.sum(axis=1)
        for k in range(batch):           # the sequential part: best-so-far and the adaptive stopping rule, in sample order
            it += 1
            if not ok[k]:
                continue
            c = int(counts[k])
            if c > best:
                best, best_mask = c, masks[k]
                w = min(max(c / n, 1e-9), 1 - 1e-9)
                need = np.log(1 - confidence) / np.log(1 - w * w)
                iters = min(iters, int(np.ceil(need)) if np.isfinite(need) else iters)
            if it >= iters:
                break
    if best < 2:
        return None
    return fit(a[best_mask], b[best_mask])


_WARNED_NO_CV2 = False


def estimate_affine_from_matches(kp0: np.ndarray, kp1: np.ndarray) -> np.ndarray:
    """MB:1431-1454: img0 -> img1 2x3 transform; identity when every estimator fails or returns non-finite values."""
    M = None
    try:
        try:
            import cv2  # type: ignore
        except ImportError:
            cv2 = None
        if cv2 is not None:
            M, _ = cv2.estimateAffinePartial2D(kp0, kp1, method=cv2.RANSAC, ransacReprojThreshold=4.0, confidence=0.999, maxIters=10000)
        else:
            global _WARNED_NO_CV2
            if not _WARNED_NO_CV2:
                logger.warning("OpenCV is not importable: PRESELECTION_AFFINE_TRANSFORM uses the built-in 2-point similarity RANSAC instead of "
                               "cv2.estimateAffinePartial2D (same model, threshold and stopping rule; tile selection can differ near rectangle borders)")
                _WARNED_NO_CV2 = True
            M = _similarity_ransac(kp0, kp1)
        if M is None:
            M = _affine_total_least_squares(kp0, kp1)
    except Exception:  # noqa: BLE001 - the reference swallows estimator failures the same way
        logger.warning("Affine estimation failed")
        M = None
    if M is None or not np.isfinite(M).all():
        M = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32)
    return np.asarray(M).astype(np.float32)


def _rects(origins: Dict[int, Tuple[int, int]], tile_size) -> Tuple[np.ndarray, np.ndarray]:
    ids = np.array(list(origins.keys()), dtype=np.int64)
    o = np.array([origins[int(t)] for t in ids], dtype=np.float32).reshape(-1, 2)
    return ids, np.concatenate([o, o + np.array([tile_size[0], tile_size[1]], np.float32)], axis=1)   # get_tile_bounding_box, float32


def select_tile_pairs_affine(kp0: np.ndarray, kp1: np.ndarray, origins0: Dict[int, Tuple[int, int]], origins1: Dict[int, Tuple[int, int]],
                             tile_size, tile_overlap: int, size1_hw, min_matches_per_tile: int = 5, M: Optional[np.ndarray] = None,
                             estimator=estimate_affine_from_matches) -> List[Tuple[int, int]]:
    """PRESELECTION_AFFINE_TRANSFORM after the preselection match (MB:1244-1333), vectorised over the tiles of image 1 and, for
    the match-count filter, over all candidate pairs.  kp0 / kp1: matched preselection keypoints in full-resolution
    coordinates; fewer than 3 of them -> the PRESELECTION vote rule (strict inequalities, ``>`` threshold, MB:1244-1258)."""
    kp0, kp1 = np.asarray(kp0, np.float32).reshape(-1, 2), np.asarray(kp1, np.float32).reshape(-1, 2)
    ids0, box0 = _rects(origins0, tile_size)
    ids1, box1 = _rects(origins1, tile_size)
    if len(kp0) < 3:
        logger.warning("Not enough matches (<3) to estimate affine transform. Falling back to standard PRESELECTION.")
        in0 = np.all(kp0[None] > box0[:, None, :2], axis=2) & np.all(kp0[None] < box0[:, None, 2:], axis=2)     # [T0, n]
        in1 = np.all(kp1[None] > box1[:, None, :2], axis=2) & np.all(kp1[None] < box1[:, None, 2:], axis=2)     # [T1, n]
        votes = in0.astype(np.int64) @ in1.astype(np.int64).T
        return sorted((int(ids0[a]), int(ids1[b])) for a, b in zip(*np.nonzero(votes > min_matches_per_tile)))
    if M is None:
        M = estimator(kp0, kp1)
    M = np.asarray(M, np.float32)
    margin = np.float32(max(2, tile_overlap))
    exp = box0 + np.array([-margin, -margin, margin, margin], np.float32)
    # transform_rectangle_with_affine (MB:1456-1470) for every tile of image 0 at once: corners in the reference's order
    corners = np.stack([exp[:, [0, 1]], exp[:, [0, 3]], exp[:, [2, 3]], exp[:, [2, 1]]], axis=1)                  # [T0, 4, 2] float32
    warped = np.stack([corners[..., 0] * M[0, 0] + corners[..., 1] * M[0, 1] + M[0, 2],
                       corners[..., 0] * M[1, 0] + corners[..., 1] * M[1, 1] + M[1, 2]], axis=2)                     # corners_h @ M.T, [T0, 4, 2]
    pred = np.concatenate([warped.min(axis=1), warped.max(axis=1)], axis=1).astype(np.float32)                     # [T0, 4]
    pred[:, [0, 2]] = np.clip(pred[:, [0, 2]], 0, size1_hw[1])
    pred[:, [1, 3]] = np.clip(pred[:, [1, 3]], 0, size1_hw[0])
    ok = (np.minimum(pred[:, None, 2], box1[None, :, 2]) > np.maximum(pred[:, None, 0], box1[None, :, 0])) & \
         (np.minimum(pred[:, None, 3], box1[None, :, 3]) > np.maximum(pred[:, None, 1], box1[None, :, 1]))          # [T0, T1]
    if ok.any() and min_matches_per_tile > 0:
        in0 = (kp0[None, :, 0] >= box0[:, None, 0]) & (kp0[None, :, 0] <= box0[:, None, 2]) & \
              (kp0[None, :, 1] >= box0[:, None, 1]) & (kp0[None, :, 1] <= box0[:, None, 3])                          # inclusive: MB:1318-1321
        in1 = (kp1[None, :, 0] >= box1[:, None, 0]) & (kp1[None, :, 0] <= box1[:, None, 2]) & \
              (kp1[None, :, 1] >= box1[:, None, 1]) & (kp1[None, :, 1] <= box1[:, None, 3])
        ok &= (in0.astype(np.int64) @ in1.astype(np.int64).T) >= min_matches_per_tile
    return sorted({(int(ids0[a]), int(ids1[b])) for a, b in zip(*np.nonzero(ok))})


class TilePreselector:
    """Device-resident PRESELECTION (MB:1054-1133): votes[t0, t1] for an image pair."""

    def __init__(self, sp_state_dict, lg_state_dict, tile_preselection_size: int = 1024, device="cuda", lib=None, cache_size: int = 64):
        self.size = int(tile_preselection_size)
        self.device = torch.device(device)
        self.lib = lib if lib is not None else capi.load()
        self._sp_sd, self._lg_sd = sp_state_dict, lg_state_dict
        self._sp: Optional[SuperPointHIP] = None
        self._sp_hw = (0, 0)
        self._lg: Optional[LightGlueHIP] = None
        self._cache: "OrderedDict[str, tuple]" = OrderedDict()
        self._cache_size = cache_size

    def _stream(self):
        return ctypes_stream(self.device)

    @staticmethod
    def _capacity() -> int:
        """keypoint slots of the preselection networks: the next power of two >= PRESELECTION_SP_CONF's max_keypoints (4000 -> 4096)"""
        return max(64, 1 << (int(PRESELECTION_SP_CONF["max_keypoints"]) - 1).bit_length())

    def _resize(self, src: torch.Tensor, h: int, w: int, linear: bool = False) -> torch.Tensor:
        H, W = src.shape
        dst = torch.empty(h, w, dtype=torch.float32, device=self.device)
        fn = self.lib.dim_op_resize_linear_f32 if linear else self.lib.dim_op_resize_area_f32
        capi.check(self.lib, fn(capi.ptr(src), H, W, capi.ptr(dst), h, w, 0, self._stream()))
        return dst

    def downsample(self, image: np.ndarray, quality: str = "HIGH") -> Tuple[torch.Tensor, float]:
        """The image-side half of tile_selection on the device: ``resize_image`` to the extraction quality (MB:1026-1034;
        utils/image.py:52-57: INTER_AREA, INTER_LINEAR when an axis is enlarged), then cv2.resize(i, size_new, INTER_AREA) to
        the preselection size (MB:1062-1069); frame2tensor's /255 happens in the extractor call."""
        if torch.is_tensor(image):      # already on the device (the tiled pipeline's extraction phase): [H, W] float32, 0..255
            src = image.to(self.device, torch.float32).contiguous()
        else:
            src = torch.as_tensor(np.ascontiguousarray(image, dtype=np.float32)).to(self.device)
        if quality != "HIGH":
            H, W = src.shape
            h, w = get_size_by_quality(quality, (H, W))
            src = self._resize(src, h, w, linear=(W < w or H < h))
        H, W = src.shape
        scale = self.size / max(W, H)
        w, h = int(round(W * scale)), int(round(H * scale))
        dst = torch.empty(h, w, dtype=torch.float32, device=self.device)
        capi.check(self.lib, self.lib.dim_op_resize_area_f32(capi.ptr(src), H, W, capi.ptr(dst), h, w, 1, self._stream()))
        return dst, scale

    def features(self, key: str, image: np.ndarray, quality: str = "HIGH"):
        """(kpts [1,cap,2], desc [1,cap,256], n [1] int32, scale) of the down-sampled image, cached by (key, quality).  ``image`` may be a
        callable returning the array (the pipeline's lazily extracted first band: not touched on a cache hit)."""
        key = (key, quality)
        if key in self._cache:
            self._cache.move_to_end(key)
            return self._cache[key]
        small, scale = self.downsample(image() if callable(image) else image, quality)
        h, w = small.shape
        if self._sp is None or h > self._sp_hw[0] or w > self._sp_hw[1]:
            self._sp_hw = (max(h, self._sp_hw[0], self.size), max(w, self._sp_hw[1], self.size))
            self._sp = SuperPointHIP(self._sp_sd, PRESELECTION_SP_CONF, max_batch=1, max_hw=self._sp_hw, capacity=self._capacity(),
                                     device=self.device, lib=self.lib)
        kp, _, de, n = self._sp.extract_batch_guarded(small[None].contiguous(), logger=logger)
        ent = (kp, de, n, scale)
        self._cache[key] = ent
        while len(self._cache) > self._cache_size:
            self._cache.popitem(last=False)
        return ent

    def match(self, f0, f1, guarded: bool = True):
        """LightGlue on two cached feature sets; image_size is absent in the reference's call (MB:1077-1079),
        so the keypoint extent is used (LGN:26-27) — computed on the device (no host read-back: the pipeline enqueues the
        preselection of all its image pairs back to back; ``guarded=False`` leaves the range-guard read to the caller's phase)."""
        if self._lg is None:
            self._lg = LightGlueHIP(self._lg_sd, PRESELECTION_LG_CONF, max_pairs=1, max_kpts=self._capacity(), device=self.device, lib=self.lib)
        kt = torch.cat([f0[0], f1[0]]).contiguous()
        dt = torch.cat([f0[1], f1[1]]).contiguous()
        nt = torch.cat([f0[2], f1[2]]).contiguous()
        st = self._extent(kt, nt)
        if guarded:
            return self._lg.match_batch_guarded(kt, dt, nt, st, n_pairs=1, logger=logger)
        return self._lg.match_batch(kt, dt, nt, st, n_pairs=1)

    def votes_device(self, key0: str, image0: np.ndarray, key1: str, image1: np.ndarray, origins0: Dict[int, Tuple[int, int]],
                     origins1: Dict[int, Tuple[int, int]], tile_size, quality: str = "HIGH", out: Optional[torch.Tensor] = None,
                     guarded: bool = True) -> torch.Tensor:
        """votes[t0, t1] (int32, sorted tile keys) left on the device; ``out``: a [n0, n1] int32 view to write into."""
        f0, f1 = self.features(key0, image0, quality), self.features(key1, image1, quality)
        o = self.match(f0, f1, guarded=guarded)
        k0, k1 = sorted(origins0), sorted(origins1)
        og0 = torch.tensor([origins0[k] for k in k0], dtype=torch.int32, device=self.device).contiguous()
        og1 = torch.tensor([origins1[k] for k in k1], dtype=torch.int32, device=self.device).contiguous()
        votes = out if out is not None else torch.empty(len(k0), len(k1), dtype=torch.int32, device=self.device)
        assert votes.is_contiguous() and votes.shape == (len(k0), len(k1))
        capi.check(self.lib, self.lib.dim_op_tile_pair_votes(
            capi.ptr(f0[0]), capi.ptr(f1[0]), capi.ptr(o["matches"]), capi.ptr(o["n_matches"]), int(o["matches"].shape[1]),
            ctypes_float(f0[3]), ctypes_float(f1[3]), capi.ptr(og0), len(k0), capi.ptr(og1), len(k1), int(tile_size[0]), int(tile_size[1]),
            capi.ptr(votes), self._stream()))
        return votes

    def _extent(self, kt: torch.Tensor, nt: torch.Tensor) -> torch.Tensor:
        """image_size stand-in of LGN:26-27 for a feature table, on the device"""
        live = torch.arange(kt.shape[1], device=self.device)[None, :, None] < nt[:, None, None]
        big = torch.finfo(torch.float32).max
        ext = 1 + torch.where(live, kt, -big).amax(1) - torch.where(live, kt, big).amin(1)
        return torch.where(nt[:, None] > 0, ext, torch.ones_like(ext)).to(torch.float32).contiguous()

    def votes_device_many(self, jobs: Sequence[tuple], tile_size, quality: str = "HIGH", pair_batch: int = 16) -> None:
        """votes of MANY image pairs (the tiled pipeline's selection phase; round 5): ``jobs`` = [(key0, image0, key1, image1, origins0, origins1, out)],
        ``out`` a contiguous [n0, n1] int32 device view.  The down-sampled features of every image are extracted once (cache), stacked into ONE
        feature table, and the pairs run through LightGlue ``pair_batch`` at a time (dim_lg_match with a pair index) instead of one call per pair —
        28 image pairs took 9 ms each as batch-1 calls (11 % of the config-5 benchmark).  No range-guard read here: the caller's phase does it."""
        if not jobs:
            return
        feats, keys = {}, []
        for k0, im0, k1, im1, *_ in jobs:
            for k, im in ((k0, im0), (k1, im1)):
                if k not in feats:
                    feats[k] = self.features(k, im, quality)
                    keys.append(k)
        idx = {k: i for i, k in enumerate(keys)}
        kt = torch.cat([feats[k][0] for k in keys]).contiguous()
        dt = torch.cat([feats[k][1] for k in keys]).contiguous()
        nt = torch.cat([feats[k][2] for k in keys]).contiguous()
        st = self._extent(kt, nt)
        nb = max(1, int(pair_batch))   # (always the full batch: a matcher sized for a short first job would be rebuilt — an allocation and a weight upload — by the next longer one)
        if getattr(self, "_lg_many", None) is None or self._lg_many_p < nb:
            self._lg_many_p = nb
            self._lg_many = LightGlueHIP(self._lg_sd, PRESELECTION_LG_CONF, max_pairs=nb, max_kpts=self._capacity(), device=self.device, lib=self.lib)
        for c0 in range(0, len(jobs), nb):
            chunk = jobs[c0:c0 + nb]
            pidx = torch.tensor([[idx[j[0]], idx[j[2]]] for j in chunk], dtype=torch.int32, device=self.device).contiguous()
            o = self._lg_many.match_batch(kt, dt, nt, st, pair_idx=pidx)
            for s, (k0, _, k1, _, origins0, origins1, votes) in enumerate(chunk):
                t0, t1 = sorted(origins0), sorted(origins1)
                og0 = torch.tensor([origins0[k] for k in t0], dtype=torch.int32, device=self.device).contiguous()
                og1 = torch.tensor([origins1[k] for k in t1], dtype=torch.int32, device=self.device).contiguous()
                assert votes.is_contiguous() and votes.shape == (len(t0), len(t1)) and votes.dtype == torch.int32
                f0, f1 = feats[k0], feats[k1]
                capi.check(self.lib, self.lib.dim_op_tile_pair_votes(
                    capi.ptr(f0[0]), capi.ptr(f1[0]), capi.ptr(o["matches"][s]), capi.ptr(o["n_matches"][s:s + 1]), int(o["matches"].shape[1]),
                    ctypes_float(f0[3]), ctypes_float(f1[3]), capi.ptr(og0), len(t0), capi.ptr(og1), len(t1), int(tile_size[0]), int(tile_size[1]),
                    capi.ptr(votes), self._stream()))

    def votes(self, key0: str, image0: np.ndarray, key1: str, image1: np.ndarray, origins0: Dict[int, Tuple[int, int]],
              origins1: Dict[int, Tuple[int, int]], tile_size, quality: str = "HIGH") -> np.ndarray:
        return self.votes_device(key0, image0, key1, image1, origins0, origins1, tile_size, quality).cpu().numpy().astype(np.int64)

    def matched_points(self, key0: str, image0: np.ndarray, key1: str, image1: np.ndarray, quality: str = "HIGH"):
        """The matched preselection keypoints scaled back to the (quality-resized) image frames: kp / scale in fp32 like numpy
        (MB:1192-1200).  The affine selection needs them on the host: 2 x (S, 2) floats leave the device."""
        f0, f1 = self.features(key0, image0, quality), self.features(key1, image1, quality)
        o = self.match(f0, f1)
        s = int(o["n_matches"][0].item())
        m = o["matches"][0, :s].cpu().numpy()
        kp0 = f0[0][0].cpu().numpy()[m[:, 0]] / np.float32(f0[3])
        kp1 = f1[0][0].cpu().numpy()[m[:, 1]] / np.float32(f1[3])
        return kp0.astype(np.float32), kp1.astype(np.float32)


def ctypes_stream(device):
    if torch.device(device).type == "cuda":
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


def ctypes_float(x):
    return ctypes.c_float(float(np.float32(x)))


def match_tile_pairs_batched(net_for, features0: dict, features1: dict, tile_pairs: Sequence[Tuple[int, int]], device,
                             pair_batch: int = 8, select_unique: bool = True) -> np.ndarray:
    """The loop of MB:414-460, batched.  ``net_for(n_kpts, n_pairs)`` returns a LightGlueHIP sized for it."""
    full = np.array([], dtype=np.int64).reshape(0, 2)
    if len(tile_pairs) == 0:
        return full
    t0s, t1s = sorted({p[0] for p in tile_pairs}), sorted({p[1] for p in tile_pairs})
    sub0 = {t: get_features_by_tile(features0, t) for t in t0s}
    sub1 = {t: get_features_by_tile(features1, t) for t in t1s}
    # A tile without keypoints cannot match anything.  (The reference would raise inside LightGlue on the
    # empty max() of LGN:26-27 and lose the whole image pair at image_matching.py:476-486; here only the
    # empty tile pairs are dropped.)
    tile_pairs = [(a, b) for a, b in tile_pairs if len(sub0[a][1]) > 0 and len(sub1[b][1]) > 0]
    if len(tile_pairs) == 0:
        return full
    items = [sub0[t] for t in t0s] + [sub1[t] for t in t1s]
    row0 = {t: i for i, t in enumerate(t0s)}
    row1 = {t: len(t0s) + i for i, t in enumerate(t1s)}
    D = features0["descriptors"].shape[0]
    cap = max(1, max(f["keypoints"].shape[0] for f, _ in items))
    kt = np.zeros((len(items), cap, 2), dtype=np.float32)
    dt = np.zeros((len(items), cap, D), dtype=np.float32)
    nt = np.zeros(len(items), dtype=np.int32)
    st = np.zeros((len(items), 2), dtype=np.float32)
    for i, (f, _) in enumerate(items):
        n = f["keypoints"].shape[0]
        kt[i, :n], dt[i, :n], nt[i] = f["keypoints"], f["descriptors"].T, n
        st[i] = np.asarray(f["image_size"], dtype=np.float32).reshape(2)
    dev = torch.device(device)
    kt_d, dt_d, nt_d, st_d = (torch.from_numpy(a).to(dev) for a in (kt, dt, nt, st))
    net = net_for(cap, pair_batch)   # (always the full batch: a handle sized for a short first job would be rebuilt — ~1 s of weight splitting, uploads and allocations — by the next longer one)
    chunks = []
    for s in range(0, len(tile_pairs), pair_batch):
        chunk = tile_pairs[s:s + pair_batch]
        pidx = torch.tensor([[row0[a], row1[b]] for a, b in chunk], dtype=torch.int32, device=dev).contiguous()
        o = net.match_batch_guarded(kt_d, dt_d, nt_d, st_d, pair_idx=pidx, n_pairs=len(chunk), logger=logger)
        cnt = o["n_matches"].cpu().numpy()
        m = o["matches"].cpu().numpy()
        for j, (a, b) in enumerate(chunk):
            c = m[j, : int(cnt[j])]
            orig = np.zeros_like(c)
            orig[:, 0] = sub0[a][1][c[:, 0]]
            orig[:, 1] = sub1[b][1][c[:, 1]]
            chunks.append(orig)
    full = np.vstack([full] + chunks)
    if select_unique:
        full, counts = np.unique(full, axis=0, return_counts=True)
        if np.any(counts > 1):
            logger.warning("Found %d duplicate matches across tile pairs", int(np.sum(counts > 1)))
    return full


def _stream_of(dev):
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if torch.device(dev).type == "cuda" else None


def _row_stride(t: torch.Tensor) -> int:
    return int(t.stride(0)) if t.shape[0] > 1 else max(1, int(t.shape[1]) if t.dim() > 1 else 1)


def device_tile_counts(lib, f: dict, n_tiles: int) -> torch.Tensor:
    """keypoints per tile of one image's merged device table (dim_op_tile_counts) -> int32 [n_tiles] on the device."""
    from . import capi
    ti = f["tile_idx"]
    counts = torch.empty(n_tiles, dtype=torch.int32, device=ti.device)
    capi.check(lib, lib.dim_op_tile_counts(capi.ptr(ti), _row_stride(ti), int(ti.numel()), n_tiles, capi.ptr(counts), _stream_of(ti.device)))
    return counts


def device_group_by_tile(lib, f: dict, row_of_tile: torch.Tensor, cap: int, kt, dt, it, nt):
    """get_features_by_tile (MB:1380-1391) for every needed tile of one image at once (dim_op_group_by_tile): tile t of the merged table ``f``
    (keypoints [N, 2], descriptors_nd [N, D], tile_idx [N]; device) fills row row_of_tile[t] (int32 device tensor, -1 = not needed) of the shared
    tables kt / dt / it / nt, keypoints in their original order."""
    import ctypes
    from . import capi
    ti = f["tile_idx"]
    n, D, T = int(ti.numel()), int(f["descriptors_nd"].shape[1]), int(row_of_tile.numel())
    if n == 0:
        return
    lib.dim_op_group_by_tile_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(int(lib.dim_op_group_by_tile_workspace_bytes(n, T)), dtype=torch.uint8, device=ti.device)
    kp, de = f["keypoints"], f["descriptors_nd"]      # (views of one packed [N][4 + D] table in the tiled pipeline: passed with their row strides, not copied)
    assert kp.stride(-1) == 1 and de.stride(-1) == 1 and ti.dtype == kp.dtype == de.dtype == torch.float32
    capi.check(lib, lib.dim_op_group_by_tile(capi.ptr(ti), _row_stride(ti), capi.ptr(kp), _row_stride(kp), capi.ptr(de), _row_stride(de), n, D, capi.ptr(row_of_tile), T,
                                             int(cap), capi.ptr(kt), capi.ptr(dt), capi.ptr(it), capi.ptr(nt), capi.ptr(ws), _stream_of(ti.device)))


def device_unique_match_rows(lib, keys: torch.Tensor, n_slots: int, cap_m: int, rows: torch.Tensor, cnt: torch.Tensor, n_full: Optional[torch.Tensor] = None):
    """np.unique(matches, axis=0) per image pair (MB:452-459) on the device (dim_op_unique_match_rows): 64-bit keys slot << 40 | idx0 << 20 | idx1
    (~0 = dead) -> rows [n_slots, cap_m, 2] (int32 or int64), cnt [n_slots]; n_full (optional) = the row counts before the cap_m cut."""
    import ctypes
    from . import capi
    n = int(keys.numel())
    lib.dim_op_unique_match_rows_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(int(lib.dim_op_unique_match_rows_workspace_bytes(ctypes.c_longlong(n))), dtype=torch.uint8, device=keys.device)
    capi.check(lib, lib.dim_op_unique_match_rows(capi.ptr(keys), ctypes.c_longlong(n), int(n_slots), int(cap_m), int(rows.dtype == torch.int64), capi.ptr(rows),
                                                 capi.ptr(cnt), capi.ptr(n_full) if n_full is not None else None, capi.ptr(ws), _stream_of(keys.device)))


def match_tile_pairs_batched_device(net_for, f0: dict, f1: dict, tile_pairs: Sequence[Tuple[int, int]], pair_batch: int = 8,
                                    select_unique: bool = True) -> torch.Tensor:
    """match_tile_pairs_batched with the feature tables ALREADY in HBM and the result left there (round 4: pipeline.TiledPairPipeline holds every
    image's merged tile table in its exchange buffer; the numpy version re-uploads 33 MB per image and image pair and unpacks on the host —
    two thirds of config 5's 108 ms per image pair).  f0 / f1: {"keypoints" [N, 2] f32, "descriptors_nd" [N, D] f32, "tile_idx" [N] f32
    (device tensors), "image_size" (2,)}.  Returns (M, 2) int64 on the device, identical to the numpy version's array (same tile
    tables, same LightGlue calls; rows unique and in np.unique(axis=0)'s lexicographic order).  Round 6: the grouping by tile, the index
    mapping and the unique are the library's own kernels (csrc/sort_ops.hip) — no torch.argsort / torch.unique on the path."""
    from . import capi
    dev = f0["keypoints"].device
    empty = torch.empty(0, 2, dtype=torch.int64, device=dev)
    if len(tile_pairs) == 0:
        return empty
    t0s, t1s = sorted({p[0] for p in tile_pairs}), sorted({p[1] for p in tile_pairs})
    lib = capi.load()
    nt0, nt1 = max(t0s) + 1, max(t1s) + 1
    cc = torch.cat([device_tile_counts(lib, f0, nt0), device_tile_counts(lib, f1, nt1)]).cpu().tolist()   # ONE host read-back per image pair (the table capacity depends on it)
    c0, c1 = cc[:nt0], cc[nt0:]
    n_of = lambda c, t: c[t] if t < len(c) else 0
    tile_pairs = [(a, b) for a, b in tile_pairs if n_of(c0, a) > 0 and n_of(c1, b) > 0]
    if len(tile_pairs) == 0:
        return empty
    cap = max(1, max([n_of(c0, t) for t in t0s] + [n_of(c1, t) for t in t1s]))
    assert cap < (1 << 20) and int(f0["keypoints"].shape[0]) < (1 << 20) and int(f1["keypoints"].shape[0]) < (1 << 20)
    D = int(f0["descriptors_nd"].shape[1])
    T = len(t0s) + len(t1s)
    kt = torch.zeros(T, cap, 2, dtype=torch.float32, device=dev)
    dt = torch.zeros(T, cap, D, dtype=torch.float32, device=dev)
    it = torch.zeros(T, cap, dtype=torch.int64, device=dev)           # tile-local slot -> index in the image's merged table
    nt = torch.zeros(T, dtype=torch.int32, device=dev)
    t0v = [t for t in t0s if n_of(c0, t) > 0]
    t1v = [t for t in t1s if n_of(c1, t) > 0]
    row0 = {t: i for i, t in enumerate(t0v)}
    row1 = {t: len(t0s) + i for i, t in enumerate(t1v)}
    for f, rows_, ntile in ((f0, row0, nt0), (f1, row1, nt1)):
        rot = torch.tensor([rows_.get(t, -1) for t in range(ntile)], dtype=torch.int32, device=dev)
        device_group_by_tile(lib, f, rot, cap, kt, dt, it, nt)
    st = torch.zeros(T, 2, dtype=torch.float32, device=dev)
    st[: len(t0s)] = torch.as_tensor(np.asarray(f0["image_size"], dtype=np.float32).reshape(2), device=dev)
    st[len(t0s):] = torch.as_tensor(np.asarray(f1["image_size"], dtype=np.float32).reshape(2), device=dev)
    net = net_for(cap, pair_batch)   # (always the full batch: a handle sized for a short first job would be rebuilt — ~1 s of weight splitting, uploads and allocations — by the next longer one)
    NK = net.nk
    keys = torch.empty(len(tile_pairs), NK, dtype=torch.int64, device=dev)
    zero_slot = torch.zeros(min(pair_batch, len(tile_pairs)), dtype=torch.int32, device=dev)
    for s in range(0, len(tile_pairs), pair_batch):
        chunk = tile_pairs[s:s + pair_batch]
        pidx = torch.tensor([[row0[a], row1[b]] for a, b in chunk], dtype=torch.int32, device=dev).contiguous()
        o = net.match_batch_guarded(kt, dt, nt, st, pair_idx=pidx, n_pairs=len(chunk), logger=logger)
        capi.check(lib, lib.dim_op_tile_match_keys(capi.ptr(o["matches"]), capi.ptr(o["n_matches"]), capi.ptr(it), capi.ptr(pidx), capi.ptr(zero_slot), len(chunk), NK, cap,
                                                   capi.ptr(keys[s:s + len(chunk)]), _stream_of(dev)))
    cap_m = len(tile_pairs) * NK
    rows = torch.empty(1, cap_m, 2, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    if select_unique:
        device_unique_match_rows(lib, keys, 1, cap_m, rows, cnt)
        return rows[0, : int(cnt.item())]
    # concatenation order, duplicates kept (select_unique False): the live keys in tile-pair order
    live = keys.reshape(-1) != -1
    k = keys.reshape(-1)[live]
    return torch.stack([(k >> 20) & 0xFFFFF, k & 0xFFFFF], 1)


def _read_band1(path: Path) -> np.ndarray:
    """rasterio ``src.read(1).astype(float32)`` (MB:1021-1024): the FIRST band, not a grey conversion."""
    try:
        import rasterio  # type: ignore
        with rasterio.open(str(path)) as src:
            return src.read(1).astype(np.float32)
    except ImportError:
        from PIL import Image  # decoder difference (rasterio/GDAL vs PIL) is outside the parity claim
        im = Image.open(str(path))
        return np.asarray(im.getchannel(0) if im.mode not in ("L", "I", "F", "I;16") else im, dtype=np.float32)


class BatchedTileMatchingMixin:
    """Overrides MatcherBase._match_by_tile.  Needs ``self._ensure_pairs(n_kpts, n_pairs)`` (LightGlueHIP),
    ``self._device``, ``self._lib`` and ``self.config['general']``."""

    tile_pair_batch = 8

    def _preselector(self) -> TilePreselector:
        if getattr(self, "_tile_preselector", None) is None:
            import os
            sp_path = self.config["general"].get("preselection_superpoint_weights") or os.environ.get("DIM_SUPERPOINT_WEIGHTS")
            lg_path = self.config["general"].get("preselection_lightglue_weights") or os.environ.get("DIM_LIGHTGLUE_WEIGHTS")
            synth = bool(self.config["general"].get("allow_synthetic_weights", False) or self.config.get("matcher", {}).get("allow_synthetic_weights", False))
            self._tile_preselector = TilePreselector(
                _weights.load_superpoint_state_dict(sp_path, allow_synthetic=synth),
                _weights.load_lightglue_state_dict(lg_path, input_dim=256, n_layers=9, allow_synthetic=synth),
                tile_preselection_size=int(self.config["general"].get("tile_preselection_size", 1024)),
                device=self._device if isinstance(self._device, (str, torch.device)) else "cuda", lib=self._lib)
        return self._tile_preselector

    def tile_selection(self, img0, img1, method: str, image0: Optional[np.ndarray] = None, image1: Optional[np.ndarray] = None):
        """tile_selection (MB:989-1140) -> sorted list of (tile0, tile1)."""
        general = self.config["general"]
        quality = getattr(general.get("quality", "HIGH"), "name", general.get("quality", "HIGH"))
        method = getattr(method, "name", method)
        i0 = image0 if image0 is not None else _read_band1(Path(img0))
        i1 = image1 if image1 is not None else _read_band1(Path(img1))
        # the tiling of the quality-resized images (MB:1026-1041): only the shapes are needed for the grid
        shape0, shape1 = get_size_by_quality(quality, i0.shape[:2]), get_size_by_quality(quality, i1.shape[:2])
        tile_size, overlap = general["tile_size"], general.get("tile_overlap", 0)
        origins0, origins1 = tile_grid(shape0, tile_size, overlap), tile_grid(shape1, tile_size, overlap)
        min_matches = int(getattr(self, "min_matches_per_tile", general.get("min_matches_per_tile", 5)))
        if method in ("PRESELECTION", "PRESELECTION_AFFINE_TRANSFORM"):
            if general.get("preselection_pipeline", "superpoint+lightglue") != "superpoint+lightglue":
                raise ValueError("Only the superpoint+lightglue preselection pipeline is built on the MI355X path")
            if method == "PRESELECTION":
                votes = self._preselector().votes(str(img0), i0, str(img1), i1, origins0, origins1, tile_size, quality)
                return select_tile_pairs(method, list(origins0), list(origins1), votes, min_matches)
            kp0, kp1 = self._preselector().matched_points(str(img0), i0, str(img1), i1, quality)
            ov = overlap if isinstance(overlap, int) else max(overlap)
            return select_tile_pairs_affine(kp0, kp1, origins0, origins1, tile_size, int(ov), shape1, min_matches)
        return select_tile_pairs(method, list(origins0), list(origins1), None, min_matches)

    @torch.no_grad()
    def _match_by_tile(self, img0, img1, features0: dict, features1: dict, method="PRESELECTION", select_unique: bool = True) -> np.ndarray:
        general = self.config["general"]
        if general.get("geometric_verification_per_tile"):
            return super()._match_by_tile(img0, img1, features0, features1, method=method, select_unique=select_unique)
        name = getattr(method, "name", method)
        tile_pairs = self.tile_selection(img0, img1, name)
        if len(tile_pairs) == 0:
            logger.debug("No tile pairs selected.")
            return np.array([], dtype=np.int64).reshape(0, 2)
        dev = self._device if isinstance(self._device, (str, torch.device)) else "cuda"
        return match_tile_pairs_batched(self._ensure_pairs, features0, features1, tile_pairs, dev, self.tile_pair_batch, select_unique)
