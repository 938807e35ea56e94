"""TEST INFRASTRUCTURE ONLY (CPU oracle) — never imported by the product package.

numpy restatement of the reference's tile-wise matching glue:

* ``get_features_by_tile``      matchers/matcher_base.py:1380-1391
* ``get_tile_bounding_box``     matchers/matcher_base.py:1401-1407
* ``points_in_rect``            matchers/matcher_base.py:1410-1412
* tile-pair selection           matchers/matcher_base.py:1042-1052 (EXHAUSTIVE / GRID) and :1124-1133 (PRESELECTION votes)
* ``MatcherBase._match_by_tile`` matchers/matcher_base.py:362-460 (sequential loop over tile pairs, index restore, np.unique)
* ``cv2.resize(..., INTER_AREA)`` as called at matchers/matcher_base.py:1068-1069; ``resize_image`` (utils/image.py:47-65)
  incl. its INTER_LINEAR branch; ``get_size_by_quality`` (constants.py:76-88)
* PRESELECTION_AFFINE_TRANSFORM selection  matchers/matcher_base.py:1244-1333, ``transform_rectangle_with_affine`` :1456-1470

Pinning: the three helper functions and the vote loop are pinned against the reference's own source
(oracle/make_golden.py executes them from matcher_base.py via ``ast`` and stores tests/golden/tile_votes.npz).
The resize is OpenCV 4.11's (uv.lock pins opencv-python 4.11.0.86) published INTER_AREA decimation
algorithm (imgproc/resize.cpp: computeResizeAreaTab / ResizeArea_Invoker / ResizeAreaFast_Invoker)
restated from its description; cv2 is absent from this container, so that part is **parity unpinned**.
"""
from __future__ import annotations

from itertools import product
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------------------------------
def get_features_by_tile(features: dict, tile_idx: int):
    if "tile_idx" not in features:
        raise KeyError("tile_idx not found in features")
    sel = features["tile_idx"] == tile_idx
    idx = np.where(sel)[0]
    return ({"keypoints": features["keypoints"][sel], "descriptors": features["descriptors"][:, sel],
             "scores": features["scores"][sel], "image_size": features["image_size"]}, idx)


def get_tile_bounding_box(origin_xy, tile_size):
    return [origin_xy[0], origin_xy[1], origin_xy[0] + tile_size[0], origin_xy[1] + tile_size[1]]


def points_in_rect(points: np.ndarray, rect) -> np.ndarray:
    return np.all(points > rect[:2], axis=1) & np.all(points < rect[2:], axis=1)


def tile_pair_votes(kp0: np.ndarray, kp1: np.ndarray, origins0: Dict[int, Tuple[int, int]], origins1: Dict[int, Tuple[int, int]],
                    tile_size) -> np.ndarray:
    """votes[t0, t1] = number of matches with kp0 inside tile t0 and kp1 inside tile t1 (MB:1124-1131)."""
    k0, k1 = sorted(origins0), sorted(origins1)
    votes = np.zeros((len(k0), len(k1)), dtype=np.int64)
    for a, t0 in enumerate(k0):
        r0 = points_in_rect(kp0, get_tile_bounding_box(origins0[t0], tile_size))
        for b, t1 in enumerate(k1):
            r1 = points_in_rect(kp1, get_tile_bounding_box(origins1[t1], tile_size))
            votes[a, b] = int(np.sum(r0 & r1))
    return votes


def select_tile_pairs(method: str, keys0: Sequence[int], keys1: Sequence[int], votes: np.ndarray = None,
                      min_matches_per_tile: int = 5) -> List[Tuple[int, int]]:
    if method == "EXHAUSTIVE":
        return sorted(product(keys0, keys1))
    if method == "GRID":
        return sorted(zip(keys0, keys1))
    if method == "PRESELECTION":
        k0, k1 = sorted(keys0), sorted(keys1)
        return sorted((k0[a], k1[b]) for a in range(len(k0)) for b in range(len(k1)) if votes[a, b] > min_matches_per_tile)
    raise ValueError(method)


def match_by_tile(features0: dict, features1: dict, tile_pairs, match_pairs: Callable[[dict, dict], np.ndarray],
                  select_unique: bool = True) -> np.ndarray:
    """MB:389-460 without the optional per-tile geometric verification."""
    full = np.array([], dtype=np.int64).reshape(0, 2)
    if len(tile_pairs) == 0:
        return full
    for t0, t1 in tile_pairs:
        f0, i0 = get_features_by_tile(features0, t0)
        f1, i1 = get_features_by_tile(features1, t1)
        c = match_pairs(f0, f1)
        orig = np.zeros_like(c)
        orig[:, 0] = i0[c[:, 0]]
        orig[:, 1] = i1[c[:, 1]]
        full = np.vstack((full, orig))
    if select_unique:
        full = np.unique(full, axis=0)
    return full


# ---------------------------------------------------------------------------------------------------
def _area_table(ssize: int, dsize: int):
    """computeResizeAreaTab: per destination index a list of (source index, fp32 weight)."""
    scale = 1.0 / (dsize / ssize)      # cv::resize: inv_scale = dsize / ssize, scale = 1. / inv_scale (not always == ssize / dsize in double)
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        taps = []
        if s1 - f1 > 1e-3:
            taps.append((s1 - 1, np.float32((s1 - f1) / cell)))
        for s in range(s1, s2):
            taps.append((s, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            taps.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        tab.append(taps)
    return tab


def resize_area(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_AREA) for a 2-D float32 image.  Decimation on both axes: the
    area tables; otherwise OpenCV's bilinear emulation of INTER_AREA (resize.cpp: sx = floor(dx * scale),
    fx = (dx + 1) - (sx + 1) * inv_scale, <= 0 -> 0 else fractional part; fx = 0 at the last source pixel)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W = img.shape
    w, h = int(size_wh[0]), int(size_wh[1])
    if h > H or w > W:
        def taps(ssize, dsize):
            inv = dsize / ssize
            scale = 1.0 / inv
            d = np.arange(dsize)
            s = np.floor(d * scale).astype(np.int64)
            f = ((d + 1) - (s + 1) * inv).astype(np.float32)
            f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
            last = s >= ssize - 1
            return np.where(last, ssize - 1, s), np.where(last, np.float32(0), f).astype(np.float32), last
        sx, fx, lastx = taps(W, w)
        sy, fy, _ = taps(H, h)
        one = np.float32(1)
        sx1 = np.minimum(sx + 1, W - 1)
        rows = (img[:, sx] * (one - fx)).astype(np.float32) + (img[:, sx1] * fx).astype(np.float32)
        rows = np.where(lastx[None, :], img[:, [W - 1]], rows).astype(np.float32)
        sy1 = np.minimum(sy + 1, H - 1)
        return ((rows[sy] * (one - fy)[:, None]).astype(np.float32) + (rows[sy1] * fy[:, None]).astype(np.float32)).astype(np.float32)
    if H % h == 0 and W % w == 0:  # ResizeAreaFast: row-major block sum, times 1/area
        sy, sx = H // h, W // w
        acc = np.zeros((h, w), dtype=np.float32)
        for y in range(sy):
            for x in range(sx):
                acc = (acc + img[y::sy, x::sx][:h, :w]).astype(np.float32)
        return (acc * np.float32(1.0 / (sx * sy))).astype(np.float32)
    xt, yt = _area_table(W, w), _area_table(H, h)
    buf = np.zeros((H, w), dtype=np.float32)
    for dx, taps in enumerate(xt):
        col = np.zeros(H, dtype=np.float32)
        for s, a in taps:
            col = (col + (img[:, s] * a).astype(np.float32)).astype(np.float32)
        buf[:, dx] = col
    out = np.zeros((h, w), dtype=np.float32)
    for dy, taps in enumerate(yt):
        row = np.zeros(w, dtype=np.float32)
        for s, b in taps:
            row = (row + (b * buf[s]).astype(np.float32)).astype(np.float32)
        out[dy] = row
    return out


def preselection_sizes(shape_hw, tile_preselection_size: int = 1024):
    """MB:1062-1067: size (w, h), scale and the rounded down-sampled size."""
    size = tuple(shape_hw[:2][::-1])
    scale = tile_preselection_size / max(size)
    return size, scale, tuple(int(round(x * scale)) for x in size)


# ---------------------------------------------------------------------------------------------------
# quality resize (utils/image.py:47-65 resize_image, constants.py:76-88 get_size_by_quality)
QUALITY_FACTOR = {"HIGHEST": 2, "HIGH": 1, "MEDIUM": 1 / 2, "LOW": 1 / 4, "LOWEST": 1 / 8}


def get_size_by_quality(quality: str, size):
    """constants.py:76-88 (pinned: make_golden.py executes the reference's function from its source)."""
    f = QUALITY_FACTOR[quality]
    return (int(size[0] * f), int(size[1] * f))


def resize_linear(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for a 2-D float32 image (OpenCV 4.11 resize.cpp restated:
    fx = (float)((dx + 0.5) * scale - 0.5), s = floor, horizontal border handling by xmin / xmax, rows clamped).
    cv2 is absent from this container: **parity unpinned**."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W = img.shape
    w, h = int(size_wh[0]), int(size_wh[1])
    one = np.float32(1)

    def taps(ssize, dsize):
        scale = 1.0 / (dsize / ssize)
        f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        return s, (f - s.astype(np.float32)).astype(np.float32)

    sx, fx = taps(W, w)
    neg = sx < 0
    sx, fx = np.where(neg, 0, sx), np.where(neg, np.float32(0), fx).astype(np.float32)
    copy = sx + 1 >= W
    sx = np.where(sx >= W - 1, W - 1, sx)
    fx = np.where(copy, np.float32(0), fx).astype(np.float32)
    sx1 = np.minimum(sx + 1, W - 1)
    rows = (img[:, sx] * (one - fx)).astype(np.float32) + (img[:, sx1] * fx).astype(np.float32)
    rows = np.where(copy[None, :], img[:, sx], rows).astype(np.float32)
    sy, fy = taps(H, h)
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    return ((rows[y0] * (one - fy)[:, None]).astype(np.float32) + (rows[y1] * fy[:, None]).astype(np.float32)).astype(np.float32)


def resize_image(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """utils/image.py:52-57 with interp="cv2_area": INTER_AREA, switched to INTER_LINEAR when either axis is enlarged."""
    h, w = img.shape[:2]
    if w < size_wh[0] or h < size_wh[1]:
        return resize_linear(img, size_wh)
    return resize_area(img, size_wh)


# ---------------------------------------------------------------------------------------------------
# PRESELECTION_AFFINE_TRANSFORM (matchers/matcher_base.py:1169-1342, helpers :1431-1470)
def transform_rectangle_with_affine(M: np.ndarray, bounds) -> np.ndarray:
    """MB:1456-1470: axis-aligned bounding box of the four transformed corners, float32."""
    xmin, ymin, xmax, ymax = bounds
    corners = np.array([[xmin, ymin], [xmin, ymax], [xmax, ymax], [xmax, ymin]], dtype=np.float32)
    warped = np.c_[corners, np.ones((4, 1), dtype=np.float32)] @ M.T
    x0, y0 = warped.min(axis=0)
    x1, y1 = warped.max(axis=0)
    return np.array([x0, y0, x1, y1], dtype=np.float32)


def affine_tile_pairs(kp0: np.ndarray, kp1: np.ndarray, M, origins0: Dict[int, Tuple[int, int]], origins1: Dict[int, Tuple[int, int]],
                      tile_size, tile_overlap: int, i1_new_size, min_matches_per_tile: int = 5) -> List[Tuple[int, int]]:
    """The selection that follows the affine estimate (MB:1244-1333), loops kept as the reference writes them.
    ``M`` None = fewer than 3 preselection matches: the PRESELECTION vote rule as fallback (MB:1244-1258)."""
    if M is None:
        votes = tile_pair_votes(kp0, kp1, origins0, origins1, tile_size)
        return select_tile_pairs("PRESELECTION", list(origins0), list(origins1), votes, min_matches_per_tile)
    rects1 = {t: np.array(get_tile_bounding_box(o, tile_size), dtype=np.float32) for t, o in origins1.items()}
    rects0 = {t: np.array(get_tile_bounding_box(o, tile_size), dtype=np.float32) for t, o in origins0.items()}
    margin = max(2, tile_overlap)
    pairs = []
    for t0, r0 in rects0.items():
        e = r0.copy()
        e[0] -= margin; e[2] += margin; e[1] -= margin; e[3] += margin
        p = transform_rectangle_with_affine(M, e)
        p[0] = np.clip(p[0], 0, i1_new_size[1]); p[2] = np.clip(p[2], 0, i1_new_size[1])
        p[1] = np.clip(p[1], 0, i1_new_size[0]); p[3] = np.clip(p[3], 0, i1_new_size[0])
        for t1, r1 in rects1.items():
            if min(p[2], r1[2]) > max(p[0], r1[0]) and min(p[3], r1[3]) > max(p[1], r1[1]):
                pairs.append((t0, int(t1)))
    if len(pairs) and min_matches_per_tile > 0:
        keep = []
        for t0, t1 in pairs:
            r0, r1 = rects0[t0], rects1[t1]
            in0 = (kp0[:, 0] >= r0[0]) & (kp0[:, 0] <= r0[2]) & (kp0[:, 1] >= r0[1]) & (kp0[:, 1] <= r0[3])
            in1 = (kp1[:, 0] >= r1[0]) & (kp1[:, 0] <= r1[2]) & (kp1[:, 1] >= r1[1]) & (kp1[:, 1] <= r1[3])
            if int(np.sum(in0 & in1)) >= min_matches_per_tile:
                keep.append((t0, t1))
        return sorted(set(keep))
    return sorted(set(pairs))
