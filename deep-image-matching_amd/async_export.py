"""Writers OFF the critical path and the end-to-end run that uses them.

In the reference every image ends with a synchronous gzip-9 float16 ``features.h5`` write (extractors/extractor_base.py:56-99)
and every pair with two ``h5`` appends (matchers/matcher_base.py:282-285,337-339) inside the hot loops, and the COLMAP
database is filled afterwards from those files (io/h5_to_db.py:44-113).  ``AsyncExporter`` keeps the same artefacts
(export.FeatureStore / MatchStore / ColmapDatabase: byte-compatible layouts) but moves the work to a writer thread:

  GPU stream ── extract / match / verify batch i+1 ───────────────────────────────────────────────►
  copy stream ── D2H of batch i's tables into pinned host buffers (event) ──►
  writer thread ──────────── wait(event) ── unpad ── fp16 / gzip / sqlite ──►

``put_*`` never blocks on the GPU: it enqueues an asynchronous device-to-host copy on a side stream into a pinned staging
buffer, records an event and hands (event, buffers, names) to the writer's queue (bounded: back-pressure instead of
unbounded host memory).  ``EndToEndRunner`` drives extraction -> matching -> device verification -> export for an image list
and reports the kernel-path and the end-to-end rates separately (SURVEY §8(e) caveat).
"""
from __future__ import annotations

import queue
import threading
import time
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import export


class AsyncExporter:
    """Background writer of features / raw matches / verified matches (+ the COLMAP database at close)."""

    def __init__(self, out_dir: Path, device="cuda", max_pending: int = 8, write_database: bool = True, camera_model: str = "simple-radial",
                 feature_workers: int = 4):
        self.out_dir = Path(out_dir)
        self.out_dir.mkdir(parents=True, exist_ok=True)
        self.device = torch.device(device)
        # The fp16 + gzip feature groups are the expensive part (~1 MB deflated per 2048-keypoint image).  h5py serialises
        # all access behind one lock, so with the real container there is one feature writer; the .npz mirror lets
        # `feature_workers` threads deflate into their own shard files in parallel (zlib releases the GIL).
        self.features = export.FeatureStore(self.out_dir / "features.h5")
        n_fw = 1 if self.features.use_h5 else max(1, int(feature_workers))
        self._fstores = [self.features] + [export.FeatureStore(self.out_dir / "features.h5", shard=i) for i in range(1, n_fw)]
        self.raw = export.MatchStore(self.out_dir / "raw_matches.h5")
        self.verified = export.MatchStore(self.out_dir / "matches.h5")
        self._write_db, self._camera_model = write_database, camera_model
        self._kpts: Dict[str, np.ndarray] = {}
        self._wh: Dict[str, Tuple[int, int]] = {}
        self._raw: Dict[Tuple[str, str], np.ndarray] = {}
        self._ver: Dict[Tuple[str, str], np.ndarray] = {}
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)      # match batches -> one writer (ordered appends + database)
        self._fq: "queue.Queue" = queue.Queue(maxsize=max_pending)     # feature batches -> the feature writers
        self._lock = threading.Lock()
        self._err: Optional[BaseException] = None
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.busy_s = 0.0            # time the writer thread spent working (hidden behind the GPU when it keeps up)
        self.n_images = self.n_pairs = 0
        self._threads = [threading.Thread(target=self._run, args=(self._q, None), name="dim-writer-matches", daemon=True)]
        self._threads += [threading.Thread(target=self._run, args=(self._fq, st), name=f"dim-writer-features{i}", daemon=True)
                          for i, st in enumerate(self._fstores)]
        for th in self._threads:
            th.start()

    # ---- producer side (GPU thread) --------------------------------------------------------------------------------
    def _stage(self, tensors: Sequence[torch.Tensor]):
        """Asynchronous D2H of device tensors into pinned buffers; returns (event or None, host tensors)."""
        if self._copy_stream is None:
            return None, [t.detach().cpu().clone() for t in tensors]
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        host = []
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self._copy_stream)
                host.append(h)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return done, host

    def put_features(self, names: Sequence[str], kpts, scores, desc, n, image_hw: Sequence[Tuple[int, int]], tile_idx=None):
        """One extracted batch: kpts [B,cap,2], scores [B,cap], desc [B,cap,D], n [B] (device) -> features.h5 groups."""
        ev, host = self._stage([kpts, scores, desc, n] + ([tile_idx] if tile_idx is not None else []))
        self._put(("features", ev, host, list(names), [tuple(hw) for hw in image_hw]), self._fq)

    def put_matches(self, pair_names: Sequence[Tuple[str, str]], matches, n_matches, mask=None):
        """One matched (and optionally verified) batch: matches [P,NK,2] int64, n_matches [P], mask [P,NK] uint8 or None."""
        ev, host = self._stage([matches, n_matches] + ([mask] if mask is not None else []))
        self._put(("matches", ev, host, list(pair_names), mask is not None), self._q)

    def _put(self, item, q):
        if self._err is not None:
            raise RuntimeError("the writer thread failed") from self._err
        q.put(item)

    # ---- writer thread ------------------------------------------------------------------------------------------
    def _run(self, q, store):
        while True:
            item = q.get()
            if item is None:
                q.task_done()
                return
            try:
                t0 = time.perf_counter()
                if item[1] is not None:
                    item[1].synchronize()       # the D2H copy of THIS batch; the GPU is already on the next one
                if item[0] == "features":
                    self._write_features(store, *item[2:])
                else:
                    self._write_matches(*item[2:])
                with self._lock:
                    self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001 - surfaced on the producer side
                self._err = e
            finally:
                q.task_done()

    def _write_features(self, store, host, names, image_hw):
        kp, sc, de, n = (h.numpy() for h in host[:4])
        ti = host[4].numpy() if len(host) > 4 else None
        for b, name in enumerate(names):
            k = int(n[b])
            feats = {"keypoints": kp[b, :k], "descriptors": np.ascontiguousarray(de[b, :k].T), "scores": sc[b, :k],
                     "tile_idx": ti[b, :k].astype(np.float32) if ti is not None else np.zeros(k, np.float32),
                     "image_size": np.array(image_hw[b])}     # (H, W), extractor_base.py:227 (Q4)
            store.add(name, feats)
            with self._lock:
                self._kpts[name] = feats["keypoints"].astype(np.float32).copy()
                self._wh[name] = (int(image_hw[b][1]), int(image_hw[b][0]))
                self.n_images += 1

    def _write_matches(self, host, pair_names, verified):
        m, n = host[0].numpy(), host[1].numpy()
        mask = host[2].numpy() if verified else None
        for p, (a, b) in enumerate(pair_names):
            s = int(n[p])
            raw = m[p, :s].copy()
            self.raw.add(a, b, raw)
            self._raw[(a, b)] = raw
            if verified and s >= 8:
                ver = raw[mask[p, :s].astype(bool)]
                self.verified.add(a, b, ver)
                self._ver[(a, b)] = ver
            self.n_pairs += 1

    # ---- shutdown ------------------------------------------------------------------------------------------------
    def close(self) -> Dict[str, float]:
        """Drains the queue, finalises the containers and writes database.db; returns the writer's statistics."""
        self._q.join(); self._fq.join()
        self._q.put(None)
        for _ in self._fstores:
            self._fq.put(None)
        for th in self._threads:
            th.join()
        if self._err is not None:
            raise RuntimeError("the writer thread failed") from self._err
        t0 = time.perf_counter()
        for st in self._fstores:
            st.close()
        self.raw.close(); self.verified.close()
        if self._write_db and self._kpts:
            names = sorted(self._kpts)
            export.export_to_colmap(self.out_dir / "database.db", names, self._wh, self._kpts, self._raw, self._ver or None,
                                    camera_model=self._camera_model)
        return {"writer_busy_s": self.busy_s, "finalise_s": time.perf_counter() - t0, "images": self.n_images, "pairs": self.n_pairs}


class EndToEndRunner:
    """extract -> (all-gather) -> match -> verify -> export for one rank, with the writers and (optionally) the verification
    off the critical path.  ``extractor`` / ``matcher``: SuperPointHIP / LightGlueHIP; ``verifier``: verify.DeviceVerifier."""

    def __init__(self, extractor, matcher, verifier=None, exporter: Optional[AsyncExporter] = None):
        self.ext, self.mat, self.ver, self.exp = extractor, matcher, verifier, exporter

    @torch.no_grad()
    def run(self, names: Sequence[str], images: torch.Tensor, pairs: torch.Tensor) -> Dict[str, float]:
        """images [n,H,W] float32 in [0,1] on the device; pairs [P,2] int32 image indices.  Returns timings (seconds)."""
        dev = images.device
        n_img, H, W = images.shape
        cap, B, PB, NK = self.ext.capacity, self.ext.max_batch, self.mat.max_pairs, self.mat.nk
        sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
        sync()
        t0 = time.perf_counter()
        kp = torch.zeros(n_img, cap, 2, device=dev); sc = torch.zeros(n_img, cap, device=dev)
        de = torch.zeros(n_img, cap, 256, device=dev); n = torch.zeros(n_img, dtype=torch.int32, device=dev)
        for s in range(0, n_img, B):
            e = min(n_img, s + B)
            k_, s_, d_, n_ = self.ext.extract_batch(images[s:e].contiguous())
            kp[s:e], sc[s:e], de[s:e], n[s:e] = k_, s_, d_, n_
            if self.exp is not None:
                self.exp.put_features(names[s:e], k_, s_, d_, n_, [(H, W)] * (e - s))
        size = torch.tensor([[float(H), float(W)]] * n_img, device=dev)
        sync()
        t1 = time.perf_counter()
        pairs_dev = pairs.to(dev, torch.int32).contiguous()
        tot_raw = tot_ver = 0
        counts = []
        for s in range(0, pairs.shape[0], PB):
            pp = pairs_dev[s:s + PB].contiguous()
            o = self.mat.match_batch(kp, de, n, size, pair_idx=pp)
            mask = None
            if self.ver is not None:
                v = self.ver.verify_batch(kp, o["matches"], o["n_matches"], pair_idx=pp)
                mask = v["mask"]
                counts.append((o["n_matches"], v["n_inliers"]))
            else:
                counts.append((o["n_matches"], None))
            if self.exp is not None:
                pn = [(names[a], names[b]) for a, b in pairs[s:s + PB].tolist()]
                self.exp.put_matches(pn, o["matches"], o["n_matches"], mask)
        sync()
        t2 = time.perf_counter()
        stats = self.exp.close() if self.exp is not None else {}
        t3 = time.perf_counter()
        for a, b in counts:
            tot_raw += int(a.sum().item())
            tot_ver += int(b.sum().item()) if b is not None else 0
        P = int(pairs.shape[0])
        return {"images": n_img, "pairs": P, "extract_s": t1 - t0, "match_verify_s": t2 - t1, "drain_s": t3 - t2,
                "kernel_path_pairs_per_s": P / (t2 - t1), "end_to_end_pairs_per_s": P / (t3 - t0), "raw_matches": tot_raw,
                "verified_matches": tot_ver, **stats}
