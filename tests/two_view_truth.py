"""Ground truth for geometric verification: synthetic two-view scenes with a KNOWN fundamental matrix and a known inlier
set.  Independent of csrc/geom_verify.hip and of its numpy restatement oracle/geom_ref.py: what is asserted here is what the
reference's estimator (cv2.findFundamentalMat, utils/geometric_verification.py:136-152 — not importable in this image) is used FOR —
separate the true correspondences from the outliers and return an F that explains them — not agreement with our own algorithm."""
from __future__ import annotations

import numpy as np


def scene(n_inliers: int, n_outliers: int, seed: int, noise_px: float = 0.4, size=(1024, 1024)):
    """-> dict(x0, x1 float32 (n, 2) noisy, c0, c1 float64 clean (inliers only, in input order), is_inlier (n,) bool, F (3, 3) true)"""
    rng = np.random.default_rng(1000 + seed)
    W, H = size
    K = np.array([[0.85 * W, 0, W / 2], [0, 0.85 * W, H / 2], [0, 0, 1.0]])
    a = rng.normal(0, 0.1, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = np.array([0.6, 0.1, 0.15]) + rng.normal(0, 0.05, 3)
    c0, c1 = [], []
    while len(c0) < n_inliers:
        X = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5), rng.uniform(3, 10)])     # general 3-D structure: no dominant plane
        p, q = K @ X, K @ (R @ X + t)
        p, q = p[:2] / p[2], q[:2] / q[2]
        if 0 <= p[0] < W and 0 <= p[1] < H and 0 <= q[0] < W and 0 <= q[1] < H:
            c0.append(p); c1.append(q)
    c0, c1 = np.asarray(c0), np.asarray(c1)
    x0 = np.concatenate([c0 + rng.normal(0, noise_px, c0.shape), rng.uniform(0, [W, H], (n_outliers, 2))])
    x1 = np.concatenate([c1 + rng.normal(0, noise_px, c1.shape), rng.uniform(0, [W, H], (n_outliers, 2))])
    is_in = np.arange(n_inliers + n_outliers) < n_inliers
    perm = rng.permutation(n_inliers + n_outliers)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    inv = np.argsort(perm)
    return {"x0": x0[perm].astype(np.float32), "x1": x1[perm].astype(np.float32), "c0": c0, "c1": c1, "clean_index": inv[:n_inliers],
            "is_inlier": is_in[perm], "F": F / np.linalg.norm(F)}


def sampson_rms(F: np.ndarray, a: np.ndarray, b: np.ndarray) -> float:
    """RMS Sampson distance (pixels) of correspondences a -> b under x_b^T F x_a = 0"""
    ha, hb = np.c_[a, np.ones(len(a))], np.c_[b, np.ones(len(b))]
    Fa, Ftb = ha @ F.T, hb @ F
    num = np.einsum("ij,ij->i", hb, Fa) ** 2
    den = Fa[:, 0] ** 2 + Fa[:, 1] ** 2 + Ftb[:, 0] ** 2 + Ftb[:, 1] ** 2
    return float(np.sqrt(np.mean(num / np.maximum(den, 1e-30))))


def score(mask: np.ndarray, F: np.ndarray, sc: dict) -> dict:
    """recall / precision of an inlier mask against the truth, and the Sampson RMS of the CLEAN true correspondences under F"""
    t = sc["is_inlier"]
    tp = int((mask & t).sum())
    return {"recall": tp / max(1, int(t.sum())), "precision": tp / max(1, int(mask.sum())),
            "sampson_rms_clean_px": sampson_rms(F, sc["c0"], sc["c1"]) if F is not None else float("nan"),
            "sampson_rms_clean_px_true_F": sampson_rms(sc["F"], sc["c0"], sc["c1"])}
