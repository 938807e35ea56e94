"""ORACLE (test infrastructure, not product code): numpy fp64 restatement of the deterministic fundamental-matrix
RANSAC of csrc/geom_verify.hip (dim_gv_fundamental), the device replacement for the reference's per-pair
``geometric_verification`` call (utils/geometric_verification.py:45-179, used at matchers/matcher_base.py:311).

The reference delegates to cv2.findFundamentalMat (USAC_MAGSAC by default) / pydegensac — third-party estimators that
are absent here and not result-identical to each other; their INTERFACE contract is what this path keeps (keypoint
pairs + pixel threshold -> F and a boolean inlier mask; fewer than 8 matches -> everything is an inlier,
geometric_verification.py:107-110).  **parity unpinned** against cv2 (no cv2 in this container); the algorithm itself
(7-point minimal solver, Sampson / symmetric epipolar scoring, normalised 8-point local optimisation) is the textbook
one (Hartley & Zisserman, Multiple View Geometry, alg. 11.4 / 11.1) and is validated against synthetic two-view
geometry with known inliers in tests/test_geom_verify_*.py.

Same sampling hash, same tie rules and the same order of refinement steps as the kernel; sums are taken in numpy's
order, so inlier sets agree with the device except for correspondences whose residual sits within rounding of the
threshold (the tests allow for that).
"""
from __future__ import annotations

import numpy as np

MASK32 = 0xFFFFFFFF


def gv_hash(seed: int, pair: int, hyp: int, k: int) -> int:
    h = (seed ^ ((pair * 0x9E3779B9) & MASK32) ^ ((hyp * 0x85EBCA6B) & MASK32) ^ ((k * 0xC2B2AE35) & MASK32)) & MASK32
    h ^= h >> 16
    h = (h * 0x7FEB352D) & MASK32
    h ^= h >> 15
    h = (h * 0x846CA68B) & MASK32
    h ^= h >> 16
    return h


def sample7(seed: int, pair: int, hyp: int, n: int):
    idx = []
    for j in range(7):
        att = 0
        while True:
            cand = (gv_hash(seed, pair, hyp, j + 7 * att) * n) >> 32
            if cand not in idx:
                idx.append(cand)
                break
            att += 1
            if att >= 8:
                return None
    return idx


def hartley(p: np.ndarray):
    c = p.mean(0)
    d = np.sqrt(((p - c) ** 2).sum(1)).mean()
    s = np.sqrt(2.0) / d if d > 0 else 1.0
    return c, s


def errors(F: np.ndarray, x0: np.ndarray, x1: np.ndarray, err_type: int = 0) -> np.ndarray:
    """Sampson distance (err_type 0) or max of the two squared point-line distances (1), in pixels^2."""
    h0 = np.concatenate([x0, np.ones((len(x0), 1))], 1)
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    l1 = h0 @ F.T        # F x0: lines in image 1
    l0 = h1 @ F          # F^T x1: lines in image 0
    e = (h1 * l1).sum(1)
    g1, g0 = l1[:, 0] ** 2 + l1[:, 1] ** 2, l0[:, 0] ** 2 + l0[:, 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        if err_type == 0:
            g = g0 + g1
            return np.where(g > 0, e * e / g, 1e300)
        return np.maximum(np.where(g1 > 0, e * e / g1, 1e300), np.where(g0 > 0, e * e / g0, 1e300))


def denormalise(Fn, c0, s0, c1, s1):
    T0 = np.array([[s0, 0, -s0 * c0[0]], [0, s0, -s0 * c0[1]], [0, 0, 1.0]])
    T1 = np.array([[s1, 0, -s1 * c1[0]], [0, s1, -s1 * c1[1]], [0, 0, 1.0]])
    return T1.T @ Fn @ T0


def _rows(x0n, x1n):
    x0, y0, x1, y1 = x0n[:, 0], x0n[:, 1], x1n[:, 0], x1n[:, 1]
    return np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones_like(x0)], 1)


def cubic_roots(c3, c2, c1, c0):
    scale = max(abs(c3), abs(c2), abs(c1), abs(c0))
    if not scale > 0:
        return []
    if abs(c3) < 1e-12 * scale:
        if abs(c2) < 1e-12 * scale:
            return [] if abs(c1) < 1e-12 * scale else [-c0 / c1]
        disc = c1 * c1 - 4 * c2 * c0
        if disc < 0:
            return []
        sq = np.sqrt(disc)
        q = -0.5 * (c1 + (sq if c1 >= 0 else -sq))
        return [q / c2, c0 / q] if q != 0 else [q / c2]
    a, b, c = c2 / c3, c1 / c3, c0 / c3
    Q, R = (a * a - 3 * b) / 9, (2 * a ** 3 - 9 * a * b + 27 * c) / 54
    Q3 = Q ** 3
    if R * R < Q3:
        th, m = np.arccos(R / np.sqrt(Q3)), -2 * np.sqrt(Q)
        return [m * np.cos(th / 3) - a / 3, m * np.cos((th + 2 * np.pi) / 3) - a / 3, m * np.cos((th - 2 * np.pi) / 3) - a / 3]
    A = -(1.0 if R >= 0 else -1.0) * np.cbrt(abs(R) + np.sqrt(R * R - Q3))
    B = Q / A if A != 0 else 0.0
    return [A + B - a / 3]


def seven_point(x0n: np.ndarray, x1n: np.ndarray):
    """Candidates (normalised frame) from 7 correspondences: Gauss-Jordan with full pivoting, closed-form cubic."""
    A = _rows(x0n, x1n).copy()
    col = list(range(9))
    for k in range(7):
        sub = np.abs(A[k:, k:])
        pr, pc = np.unravel_index(int(np.argmax(sub)), sub.shape)   # first maximum in row-major order, like the kernel's scan
        if not sub[pr, pc] > 1e-12:
            return []
        pr, pc = pr + k, pc + k
        if pr != k:
            A[[k, pr]] = A[[pr, k]]
        if pc != k:
            A[:, [k, pc]] = A[:, [pc, k]]
            col[k], col[pc] = col[pc], col[k]
        A[k] = A[k] * (1.0 / A[k, k])
        for i in range(7):
            if i != k and A[i, k] != 0:
                A[i] = A[i] - A[i, k] * A[k]
    F1, F2 = np.zeros(9), np.zeros(9)
    F1[col[7]], F2[col[8]] = 1.0, 1.0
    for k in range(7):
        F1[col[k]], F2[col[k]] = -A[k, 7], -A[k, 8]
    pv = [np.linalg.det((a * F1 + (1 - a) * F2).reshape(3, 3)) for a in (0.0, 1.0, -1.0, 2.0)]
    c0 = pv[0]
    c2 = 0.5 * (pv[1] + pv[2]) - c0
    s = 0.5 * (pv[1] - pv[2])
    t = pv[3] - 4 * c2 - c0
    c3 = (t - 2 * s) / 6
    c1 = s - c3
    return [(r * F1 + (1 - r) * F2).reshape(3, 3) for r in cubic_roots(c3, c2, c1, c0)]


def eight_point_ls(x0n, x1n):
    """Normalised 8-point least squares + rank-2 enforcement (the kernel's local-optimisation step)."""
    M = _rows(x0n, x1n)
    w, V = np.linalg.eigh(M.T @ M)
    Fn = V[:, 0].reshape(3, 3)
    w3, V3 = np.linalg.eigh(Fn.T @ Fn)
    v3 = V3[:, 0]
    return Fn - np.outer(Fn @ v3, v3)


def fundamental_ransac(x0: np.ndarray, x1: np.ndarray, threshold: float, iters: int = 1024, err_type: int = 0, seed: int = 0,
                       pair: int = 0):
    """-> (F 3x3 or None, inlier mask bool [n], n_inliers, best_hypothesis_id).  x0, x1: (n, 2) float32 pixel coordinates."""
    x0, x1 = np.asarray(x0, np.float32).astype(np.float64), np.asarray(x1, np.float32).astype(np.float64)
    n = len(x0)
    if n < 8:
        return None, np.ones(n, bool), n, -1
    thr2 = float(threshold) ** 2
    c0, s0 = hartley(x0)
    c1, s1 = hartley(x1)
    x0n, x1n = (x0 - c0) * s0, (x1 - c1) * s1
    best = (-1, 0x7FFFFFFF, None)
    for hyp in range(iters):
        idx = sample7(seed, pair, hyp, n)
        if idx is None:
            continue
        for r, Fn in enumerate(seven_point(x0n[idx], x1n[idx])):
            F = denormalise(Fn, c0, s0, c1, s1)
            cnt = int((errors(F, x0, x1, err_type) <= thr2).sum())
            hid = hyp * 3 + r
            if cnt > best[0] or (cnt == best[0] and hid < best[1]):
                best = (cnt, hid, F)
    cnt, hid, F = best
    if F is None:
        return None, np.zeros(n, bool), 0, -1
    for _ in range(2):
        if cnt < 8:
            break
        inl = errors(F, x0, x1, err_type) <= thr2
        F2 = denormalise(eight_point_ls(x0n[inl], x1n[inl]), c0, s0, c1, s1)
        c2 = int((errors(F2, x0, x1, err_type) <= thr2).sum())
        if c2 >= cnt:
            cnt, F = c2, F2
    mask = errors(F, x0, x1, err_type) <= thr2
    nrm = np.linalg.norm(F)
    F = F / F[2, 2] if abs(F[2, 2]) > 1e-12 * nrm else F / nrm
    return F, mask, int(mask.sum()), hid


def synthetic_two_view(n_inliers: int, n_outliers: int, seed: int = 0, noise_px: float = 0.3, size=(1024, 1024)):
    """Two pinhole views of random 3-D points + uniform outliers.  -> (x0, x1 float32 (n,2), is_inlier bool, F_true)."""
    rng = np.random.default_rng(seed)
    W, H = size
    K = np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1.0]])
    ang = rng.normal(0, 0.08, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = np.array([0.5, 0.05, 0.1]) + rng.normal(0, 0.05, 3)
    pts, x0, x1 = [], [], []
    while len(x0) < n_inliers:
        X = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(3, 9)])
        a, b = K @ X, K @ (R @ X + t)
        a, b = a[:2] / a[2], b[:2] / b[2]
        if 0 <= a[0] < W and 0 <= a[1] < H and 0 <= b[0] < W and 0 <= b[1] < H:
            x0.append(a + rng.normal(0, noise_px, 2))
            x1.append(b + rng.normal(0, noise_px, 2))
    for _ in range(n_outliers):
        x0.append(rng.uniform(0, [W, H]))
        x1.append(rng.uniform(0, [W, H]))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    is_in = np.zeros(n_inliers + n_outliers, bool)
    is_in[:n_inliers] = True
    perm = rng.permutation(n_inliers + n_outliers)
    return np.asarray(x0, np.float32)[perm], np.asarray(x1, np.float32)[perm], is_in[perm], F / F[2, 2]
