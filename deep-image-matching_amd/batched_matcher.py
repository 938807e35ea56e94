"""Batched replacement for the two hot loops of ``ImageMatcher`` (image_matching.py:413-494).

The reference calls ``self._extractor.extract(img)`` once per image and ``self._matcher.match(...)`` once per pair: batch 1,
one host synchronisation and one h5 transaction per call — its loop would see ≈ 250 pairs/s of the ≈ 500 the kernels
deliver.  ``BatchedImageMatcher`` runs the same two phases through the batched entry points the plugins
already own (``_ensure_batch`` -> ``dim_sp_extract`` with B images, ``_ensure_pairs`` -> ``dim_lg_match`` with P pairs and a
pair-index table, ``dim_gv_fundamental`` for the verification) and writes the same artefacts with the same rules:

  extract_features(images)      -> features.h5   (float16 groups, tile_idx = 0, image_size = (H, W); extractor_base.py:205-232)
  match_pairs(feature_path, pairs) -> raw_matches.h5 + matches.h5 (pairs with < 8 raw matches are skipped, then
                                   min_inliers_per_pair / min_inlier_ratio_per_pair; matcher_base.py:282-339)

Features are re-read from the container before matching, exactly like ``MatcherBase.match`` does (the float16 round trip of
Q6 is part of the reference's behaviour), so the matches equal what the per-pair loop produces on the same files.
Tiling and quality resizing are not handled here (the per-image hooks of the plugins cover them).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import export
from .verify import DeviceVerifier, QUALITY_GV_SCALE, apply_reference_filters

logger = logging.getLogger("dim")


def default_image_loader(path: Path, grayscale: bool = True) -> np.ndarray:
    """ExtractorBase.extract's image read (extractor_base.py:186-202): rasterio bands -> (H, W[, C]) -> grey -> float32.
    Without rasterio / cv2 (this container) PIL decodes and the 8-bit BGR2GRAY fixed-point formula is restated; decoder
    differences are outside the parity claim."""
    try:
        import rasterio  # type: ignore

        with rasterio.open(str(path)) as src:
            img = np.transpose(src.read(), (1, 2, 0))
    except ImportError:
        from PIL import Image

        img = np.asarray(Image.open(str(path)))
        img = img[..., None] if img.ndim == 2 else img
    if img.shape[2] == 1:
        img = img[:, :, 0]
    if grayscale and img.ndim == 3:
        a = img.astype(np.int64)   # cv2.cvtColor(image, COLOR_BGR2GRAY) on uint8: (B*1868 + G*9617 + R*4899 + 8192) >> 14
        img = ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(img.dtype)
    return img.astype(np.float32)


class BatchedImageMatcher:
    """extractor / matcher: the plugin instances of plugins.py (SuperPointExtractor, LightGlueMatcher)."""

    def __init__(self, extractor, matcher, output_dir: Path, image_batch: int = 16, pair_batch: int = 16,
                 loader: Optional[Callable[[Path], np.ndarray]] = None, verify: bool = True, gv_iters: int = 2048):
        self.ext, self.mat = extractor, matcher
        self.out = Path(output_dir)
        self.out.mkdir(parents=True, exist_ok=True)
        self.image_batch, self.pair_batch = int(image_batch), int(pair_batch)
        self.loader = loader or (lambda p: default_image_loader(p, getattr(extractor, "grayscale", True)))
        general = getattr(matcher, "config", {}).get("general", {})
        q = general.get("quality", "HIGH")
        thr = float(general.get("gv_threshold", 4)) * QUALITY_GV_SCALE.get(getattr(q, "name", str(q)).upper(), 1.0)
        self.min_inliers = int(general.get("min_inliers_per_pair", 15))
        self.min_ratio = float(general.get("min_inlier_ratio_per_pair", 0.25))
        gv = general.get("geom_verification", "MAGSAC")
        self.verify = verify and getattr(gv, "name", str(gv)).upper() != "NONE"
        dev = self.ext._device
        self._verifier = DeviceVerifier(threshold=thr, iters=gv_iters, device=dev, lib=self.ext._lib) if self.verify else None

    # ---- phase 1: image_matching.py:413-436 ------------------------------------------------------------------------
    @torch.no_grad()
    def extract_features(self, images: Sequence[Path]) -> Path:
        feature_path = self.out / "features.h5"
        store = export.FeatureStore(feature_path)
        # Images are bucketed by shape and a bucket is extracted as soon as it holds ``image_batch`` images, so the host
        # keeps at most image_batch decoded images per distinct shape (the reference holds one; decoding everything up
        # front would be O(dataset) of float32 in host memory).
        by_shape: Dict[Tuple[int, ...], List[Tuple[Path, np.ndarray]]] = {}

        def flush(shape, chunk):
            H, W = shape[:2]
            net = self.ext._ensure_batch(H, W, self.image_batch)
            # _frame2tensor's /255 as the reference computes it — numpy's true division on the host; a device-side `tensor / 255.0` is a
            # multiplication by the rounded reciprocal and differs in the last bit of some pixels (found on the real photographs of config 1:
            # the hooks' and the batched path's float16 descriptors differed in a few elements)
            stack = torch.from_numpy(np.ascontiguousarray(np.stack([im for _, im in chunk]) / 255.0, dtype=np.float32)).to(net.device)
            run = getattr(net, "extract_batch_guarded", net.extract_batch)
            kp, sc, de, n = run(stack.contiguous())
            if hasattr(self.ext, "_regrow") and self.ext._regrow(net, len(chunk)):
                net = self.ext._ensure_batch(H, W, self.image_batch)
                kp, sc, de, n = getattr(net, "extract_batch_guarded", net.extract_batch)(stack.contiguous())
            kp, sc, de, n = kp.cpu().numpy(), sc.cpu().numpy(), de.cpu().numpy(), n.cpu().numpy()
            for j, (p, _) in enumerate(chunk):
                k = int(n[j])
                write(p.name, {"keypoints": kp[j, :k].copy(), "descriptors": np.ascontiguousarray(de[j, :k].T), "scores": sc[j, :k].copy(),
                               "tile_idx": np.zeros(k, np.float32), "image_size": np.array((H, W))})

        # The host side of a job is as long as its device side (one image: ~10 - 50 ms of JPEG decode + grey conversion, ~40 ms of float16 conversion +
        # deflate of its features; ~1 ms of kernels): images are decoded by a small thread pool a window ahead of the extraction (PIL and numpy release
        # the GIL; order preserved, at most 2 image_batch decoded images wait), and the feature store is written by background threads
        # (zlib releases the GIL too) — the stored datasets are the same bytes as with the serial loop.
        import collections
        import os
        from concurrent.futures import ThreadPoolExecutor
        images = [Path(p) for p in images]
        pending_writes = collections.deque()
        # (h5py serialises on its global lock: one writer; the .npz mirror has shard files for exactly this — FeatureStore.read sees their union)
        n_writers = 1 if store.use_h5 else max(1, min(4, (os.cpu_count() or 2) // 2))
        if not store.use_h5:
            for stale in feature_path.parent.glob(feature_path.stem + ".shard*.npz"):      # shards of an earlier run in this directory would be read as part of this one
                stale.unlink()
        stores = [store] + [export.FeatureStore(feature_path, shard=k) for k in range(1, n_writers)]
        writers = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"dim-features{k}") for k in range(n_writers)]
        turn = [0]

        def write(name, feats):      # flush() hands every image's features here: round-robin over the writer threads, each with its own shard
            k = turn[0] % n_writers
            turn[0] += 1
            pending_writes.append(writers[k].submit(stores[k].add, name, feats))

        try:
            with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1)), thread_name_prefix="dim-decode") as decoders:
                window, ahead = 2 * self.image_batch, collections.deque()
                it = iter(images)
                for p in it:
                    ahead.append((p, decoders.submit(self.loader, p)))
                    if len(ahead) >= window:
                        break
                while ahead:
                    p, fut = ahead.popleft()
                    img = fut.result()
                    nxt = next(it, None)
                    if nxt is not None:
                        ahead.append((nxt, decoders.submit(self.loader, nxt)))
                    bucket = by_shape.setdefault(img.shape, [])
                    bucket.append((p, img))
                    if len(bucket) >= self.image_batch:
                        flush(img.shape, bucket)
                        by_shape[img.shape] = []
                    while len(pending_writes) > 4 * self.image_batch:      # bounded: the writer is at most a few batches behind
                        pending_writes.popleft().result()
                for shape, bucket in by_shape.items():
                    if bucket:
                        flush(shape, bucket)
                while pending_writes:
                    pending_writes.popleft().result()                      # (re-raises a writer error here)
        finally:
            for w in writers:
                w.shutdown(wait=True)
            for st_ in stores:
                st_.close()
        return feature_path

    # ---- phase 2: image_matching.py:438-494 ------------------------------------------------------------------------
    @torch.no_grad()
    def match_pairs(self, feature_path: Path, pairs: Sequence[Tuple[str, str]]) -> Path:
        feature_path = Path(feature_path)
        matches_path = feature_path.parent / "matches.h5"
        raw_store, ver_store = export.MatchStore(feature_path.parent / "raw_matches.h5"), export.MatchStore(matches_path)
        names = sorted({Path(n).name for pr in pairs for n in pr})
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1)), thread_name_prefix="dim-read") as pool:      # (inflate releases the GIL: ~9 ms per image serially)
            feats = dict(zip(names, pool.map(lambda n: export.FeatureStore.read(feature_path, n), names)))      # the float16 round trip of the reference
        slot = {n: i for i, n in enumerate(names)}
        cap = max(1, max(f["keypoints"].shape[0] for f in feats.values()))
        D = next(iter(feats.values()))["descriptors"].shape[0]
        kt, dt = np.zeros((len(names), cap, 2), np.float32), np.zeros((len(names), cap, D), np.float32)
        nt, st = np.zeros(len(names), np.int32), np.zeros((len(names), 2), np.float32)
        for n, i in slot.items():
            f = feats[n]
            k = f["keypoints"].shape[0]
            kt[i, :k], dt[i, :k], nt[i], st[i] = f["keypoints"], f["descriptors"].T, k, f["image_size"].astype(np.float32)
        net = self.mat._ensure_pairs(cap, self.pair_batch)
        dev = net.device
        if self._verifier is not None and net.nk > 4096:
            # dim_gv_fundamental stages a pair's correspondences in LDS: at most 4096 matches per pair
            raise ValueError(f"device verification handles at most 4096 keypoints per image (this run needs {net.nk}); lower "
                             "max_keypoints, or construct BatchedImageMatcher(verify=False) and verify with verify.HostVerifierPool")
        kt_d, dt_d, nt_d, st_d = (torch.from_numpy(a).to(dev) for a in (kt, dt, nt, st))
        for s in range(0, len(pairs), self.pair_batch):
            chunk = [(Path(a).name, Path(b).name) for a, b in pairs[s:s + self.pair_batch]]
            pidx = torch.tensor([[slot[a], slot[b]] for a, b in chunk], dtype=torch.int32, device=dev).contiguous()
            o = net.match_batch_guarded(kt_d, dt_d, nt_d, st_d, pair_idx=pidx, n_pairs=len(chunk))
            mask = None
            if self._verifier is not None:
                mask = self._verifier.verify_batch(kt_d, o["matches"], o["n_matches"], pair_idx=pidx)["mask"].cpu().numpy()
            cnt, m = o["n_matches"].cpu().numpy(), o["matches"].cpu().numpy()
            for j, (a, b) in enumerate(chunk):
                raw = m[j, : int(cnt[j])].copy()
                raw_store.add(a, b, raw)
                keep = apply_reference_filters(raw, mask[j, : len(raw)] if mask is not None else np.ones(len(raw), bool),
                                               self.min_inliers, self.min_ratio)
                if keep is not None:
                    ver_store.add(a, b, keep)
        raw_store.close(); ver_store.close()
        return matches_path
