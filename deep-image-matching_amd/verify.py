"""Geometric verification at matching rate.

The reference verifies every pair on the host right after ``_match_pairs`` (matchers/matcher_base.py:298-339:
``geometric_verification`` -> cv2.findFundamentalMat / pydegensac, one call per pair, ~ms each) — at several hundred
pairs per second per GPU that serial host call is the end-to-end bottleneck.  Two replacements:

* ``DeviceVerifier``  — the batched fundamental-matrix RANSAC of csrc/geom_verify.hip (``dim_gv_fundamental``) straight
  on the device tables ``dim_lg_match`` produced: no host round trip, one launch pair per batch.  Deterministic; restated by
  oracle/geom_ref.py.  Same interface contract as the reference (pixel threshold -> F + boolean inlier mask; < 8 matches
  -> all inliers), not result-identical to cv2's MAGSAC.
* ``HostVerifierPool`` — the REFERENCE's own estimator (cv2.findFundamentalMat with the reference's method table, or
  pydegensac) on a thread pool, fed from the device tables, for users who need cv2's exact inlier sets: the calls release
  the GIL, so ``workers`` pairs are verified concurrently while the GPU matches the next batch.  Needs cv2 (absent in the
  build container: import-guarded, covered by a fake-estimator test only).

``apply_reference_filters`` restates the accept / reject rules that follow the estimator in MatcherBase.match
(min_inliers_per_pair, min_inlier_ratio_per_pair, "fewer than 8 raw matches -> skip the pair").
"""
from __future__ import annotations

import ctypes
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi

ERROR_TYPES = {"sampson": 0, "symmetric_epipolar": 1}
QUALITY_GV_SCALE = {"HIGHEST": 1.0, "HIGH": 1.0, "MEDIUM": 1.5, "LOW": 2.0, "LOWEST": 3.0}   # matcher_base.py:296-302


class DeviceVerifier:
    """Batched F-matrix RANSAC on the match tables of ``LightGlueHIP.match_batch`` (device in, device out, no sync)."""

    def __init__(self, threshold: float = 4.0, iters: int = 2048, error_type: str = "sampson", seed: int = 0, device="cuda", lib=None):
        self.lib = lib if lib is not None else capi.load()
        self.device = torch.device(device)
        self.threshold, self.iters, self.seed = float(threshold), int(iters), int(seed)
        if error_type not in ERROR_TYPES:
            raise ValueError(f"error_type must be one of {sorted(ERROR_TYPES)}")
        self.error_type = ERROR_TYPES[error_type]
        self.lib.dim_gv_scratch_bytes.restype = ctypes.c_size_t
        self._scratch: Optional[torch.Tensor] = None

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    @torch.no_grad()
    def verify_batch(self, kpts_tab: torch.Tensor, matches: torch.Tensor, n_matches: torch.Tensor, pair_idx: Optional[torch.Tensor] = None,
                     out=None):
        """kpts_tab [n_img, cap, 2] f32; matches [P, NK, 2] int64 and n_matches [P] int32 as written by dim_lg_match;
        pair_idx [P, 2] int32 or None (pair p = slots 2p, 2p+1).  Returns device tensors
        {"mask" [P, NK] uint8, "n_inliers" [P] int32, "F" [P, 3, 3] float64}."""
        assert kpts_tab.dtype == torch.float32 and matches.dtype == torch.int64 and n_matches.dtype == torch.int32
        assert kpts_tab.is_contiguous() and matches.is_contiguous() and n_matches.is_contiguous()
        P, NK = matches.shape[0], matches.shape[1]
        dev = matches.device
        need = int(self.lib.dim_gv_scratch_bytes(P))
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        if out is None:
            out = {"mask": torch.empty(P, NK, dtype=torch.uint8, device=dev), "n_inliers": torch.empty(P, dtype=torch.int32, device=dev),
                   "F": torch.empty(P, 3, 3, dtype=torch.float64, device=dev)}
        ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
        with ctx:
            capi.check(self.lib, self.lib.dim_gv_fundamental(
                capi.ptr(kpts_tab), int(kpts_tab.shape[1]), capi.ptr(pair_idx), capi.ptr(matches), capi.ptr(n_matches), int(NK), int(P),
                ctypes.c_double(self.threshold), int(self.iters), int(self.error_type), ctypes.c_uint(self.seed & 0xFFFFFFFF),
                capi.ptr(self._scratch), ctypes.c_size_t(self._scratch.numel()), capi.ptr(out["mask"]), capi.ptr(out["n_inliers"]),
                capi.ptr(out["F"]), self._stream()))
        return out

    def verify_pair(self, kpts0: np.ndarray, kpts1: np.ndarray, matches: np.ndarray) -> Tuple[Optional[np.ndarray], np.ndarray]:
        """The reference's call shape for ONE pair (geometric_verification(kpts0=..., kpts1=...)): -> (F or None, inlMask)."""
        n0, n1, S = len(kpts0), len(kpts1), len(matches)
        cap = max(n0, n1, 1)
        kt = torch.zeros(2, cap, 2, dtype=torch.float32)
        kt[0, :n0], kt[1, :n1] = torch.as_tensor(kpts0, dtype=torch.float32), torch.as_tensor(kpts1, dtype=torch.float32)
        mt = torch.zeros(1, max(S, 1), 2, dtype=torch.int64)
        mt[0, :S] = torch.as_tensor(np.asarray(matches), dtype=torch.int64)
        o = self.verify_batch(kt.to(self.device), mt.to(self.device), torch.tensor([S], dtype=torch.int32, device=self.device))
        mask = o["mask"][0, :S].cpu().numpy().astype(bool)
        return (o["F"][0].cpu().numpy() if S >= 8 else None), mask


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _reference_estimator(method: str, threshold: float, confidence: float, max_iters: int) -> Callable:
    """cv2 / pydegensac exactly as utils/geometric_verification.py:120-171 calls them."""
    import importlib

    cv2 = importlib.import_module("cv2")   # ImportError where OpenCV is absent
    table = {"MAGSAC": "USAC_MAGSAC", "RANSAC": "RANSAC", "LMEDS": "LMEDS", "RHO": "RHO", "USAC_DEFAULT": "USAC_DEFAULT",
             "USAC_PARALLEL": "USAC_PARALLEL", "USAC_FM_8PTS": "USAC_FM_8PTS", "USAC_FAST": "USAC_FAST", "USAC_ACCURATE": "USAC_ACCURATE",
             "USAC_PROSAC": "USAC_PROSAC", "USAC_MAGSAC": "USAC_MAGSAC"}
    if method.upper() == "PYDEGENSAC":
        pyd = importlib.import_module("pydegensac")
        return lambda a, b: pyd.findFundamentalMatrix(a, b, px_th=threshold, conf=confidence, max_iters=max_iters, laf_consistensy_coef=-1.0,
                                                      error_type="sampson", symmetric_error_check=True, enable_degeneracy_check=True)
    flag = getattr(cv2, table[method.upper()])

    def run(a, b):
        F, inl = cv2.findFundamentalMat(a, b, flag, threshold, confidence, max_iters)
        return F, (np.asarray(inl) > 0).reshape(-1) if inl is not None else np.ones(len(a), bool)

    return run


class HostVerifierPool:
    """The reference's estimator on ``workers`` threads: submit() returns futures, so verification of batch i overlaps the
    GPU work of batch i+1 (cv2 releases the GIL inside findFundamentalMat)."""

    def __init__(self, method: str = "MAGSAC", threshold: float = 4.0, confidence: float = 0.99999, max_iters: int = 10000, workers: int = 16,
                 estimator: Optional[Callable] = None):
        self._est = estimator if estimator is not None else _reference_estimator(method, threshold, confidence, max_iters)
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="dim-gv")

    def _one(self, k0, k1, m):
        if len(m) < 8:   # geometric_verification.py:107-110
            return None, np.ones(len(m), bool)
        return self._est(k0[m[:, 0]], k1[m[:, 1]])

    def submit(self, kpts0: np.ndarray, kpts1: np.ndarray, matches: np.ndarray):
        return self._pool.submit(self._one, kpts0, kpts1, matches)

    def shutdown(self):
        self._pool.shutdown(wait=True)


def apply_reference_filters(matches: np.ndarray, mask: np.ndarray, min_inliers_per_pair: int = 15, min_inlier_ratio_per_pair: float = 0.25):
    """MatcherBase.match after the estimator (matcher_base.py:287-334): None when the pair is dropped, else the inlier rows."""
    if len(matches) < 8:
        return None
    n_in = int(np.sum(mask))
    if n_in < min_inliers_per_pair or n_in / len(matches) < min_inlier_ratio_per_pair:
        return None
    return matches[np.asarray(mask, bool)]
