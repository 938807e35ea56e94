"""Writers OFF the critical path and the end-to-end run that uses them.

In the reference every image ends with a synchronous gzip-9 float16 ``features.h5`` write (extractors/extractor_base.py:56-99)
and every pair with two ``h5`` appends (matchers/matcher_base.py:282-285,337-339) inside the hot loops, and the COLMAP
database is filled afterwards from those files (io/h5_to_db.py:44-113).  ``AsyncExporter`` keeps the same artefacts
(export.FeatureStore / MatchStore / ColmapDatabase: the same groups, datasets, dtypes and blobs) with the work split so that
the GPU never waits and the host only does what cannot be done in HBM:

  GPU stream   ── extract / match / verify batch i+1 ─────────────────────────────────────────────────────────►
               └ dim_op_pack_features_f16: fp32 -> fp16, (N, D) -> (D, N), un-padding to the live counts  (csrc/export_ops.hip)
               └ dim_op_filter_matches:    < 8 matches / min_inliers / min_inlier_ratio rules + inlier compaction
  copy stream  ── ONE D2H of the packed fp16 slots (half the bytes of the fp32 tables) into a pinned ring buffer ──►
  worker pool  ────────── wait(event) ── per image: deflate-9 the dataset byte images as they lie in the slot ──►
  match writer ────────── wait(event) ── raw / verified appends, incremental database.db rows ──►

``put_*`` never blocks on the GPU: it enqueues kernels and an asynchronous copy, records an event and hands (event, pinned
buffer, names) to the queues.  The pinned buffers form a ring that is allocated once (hipHostMalloc per batch was part of the
round-2 slowdown) and recycled when the last image of a batch has been written; a producer that runs ahead of the writers by
more than ``max_pending`` batches blocks on the ring (bounded host memory).

The deflate pool is sized from the deflate rate measured at construction (gzip-9 of fp16 descriptors runs at ≈ 25-30 MB/s per
core, i.e. ≈ 37 ms per 2048-keypoint SuperPoint image).  With the ``.npz`` mirror every worker appends to its own shard; with
h5py the workers pre-compress (zlib releases the GIL) and one lock-protected writer stores the finished chunk with
``write_direct_chunk`` — HDF5 then only copies bytes, so the single-writer constraint of h5py does not serialise the deflate.

``EndToEndRunner`` drives extraction -> matching -> device verification -> export for an image list under the fp16x3 range guard
and reports the kernel-path and the end-to-end rates separately (SURVEY §8(e) caveat).
"""
from __future__ import annotations

import ctypes
import os
import queue
import threading
import time
import zlib
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi, export


def measure_deflate_rate(level: int = 9, nbytes: int = 1 << 18) -> float:
    """MB/s of zlib level-``level`` on float16 unit-norm descriptor bytes (what features.h5 mostly holds), one core."""
    rng = np.random.default_rng(0)
    d = rng.standard_normal((256, nbytes // 512)).astype(np.float32)
    buf = (d / np.linalg.norm(d, axis=0)).astype(np.float16).tobytes()
    t0 = time.perf_counter()
    zlib.compress(buf, level)
    return len(buf) / 1e6 / max(time.perf_counter() - t0, 1e-6)


class _Ring:
    """A pool of equally sized host staging buffers (pinned when a GPU is used) with blocking acquire."""

    def __init__(self, count: int, pinned: bool):
        self._count, self._pinned = count, pinned
        self._free: "queue.Queue" = queue.Queue()
        self._made = 0
        self._lock = threading.Lock()

    def acquire(self, shapes_dtypes, failed=None):
        """-> list of host tensors matching ``shapes_dtypes`` = [(shape, dtype), ...] (re-used when the sizes still fit).
        Blocks while all ``count`` buffers are with the writers; ``failed()`` (the exporter's error check, raises) is polled
        meanwhile, so a producer that is ahead of a writer which died cannot wait forever."""
        with self._lock:
            make = self._free.empty() and self._made < self._count
            if make:
                self._made += 1
        bufs = None
        while not make:
            try:
                bufs = self._free.get(timeout=0.5)
                break
            except queue.Empty:
                if failed is not None:
                    failed()
        out = []
        for i, (shape, dtype) in enumerate(shapes_dtypes):
            need = int(np.prod(shape))
            old = bufs[i] if bufs is not None and i < len(bufs) else None
            if old is None or old.dtype != dtype or old.numel() < need:
                old = torch.empty(max(need, 1), dtype=dtype, pin_memory=self._pinned)
            out.append(old)
        return out

    def release(self, bufs):
        self._free.put(bufs)


class AsyncExporter:
    """Background writer of features / raw matches / verified matches and the COLMAP database.

    ``image_names``: when the full image list is known up front, image ids are pre-assigned (sorted order, as
    io/h5_to_db.py:add_keypoints walks the sorted image directory) and database rows are inserted while the run is in flight;
    otherwise the database is written at ``close`` from the tables collected on the way."""

    def __init__(self, out_dir: Path, device="cuda", max_pending: int = 8, write_database: bool = True, camera_model: str = "simple-radial",
                 feature_workers: Optional[int] = None, min_inliers_per_pair: int = 15, min_inlier_ratio_per_pair: float = 0.25,
                 image_names: Optional[Sequence[str]] = None, lib=None, expected_images_per_s: float = 150.0):
        self.out_dir = Path(out_dir)
        self.out_dir.mkdir(parents=True, exist_ok=True)
        self.device = torch.device(device)
        self.lib = lib if lib is not None else capi.load()
        self.lib.dim_pack_features_slot_halves.restype = ctypes.c_size_t
        self.min_inliers, self.min_ratio = int(min_inliers_per_pair), float(min_inlier_ratio_per_pair)
        self.deflate_mb_per_s = measure_deflate_rate()
        if feature_workers is None:   # enough cores to deflate `expected_images_per_s` images of ~1.06 MB each, within the machine
            want = int(np.ceil(expected_images_per_s * 1.06 / self.deflate_mb_per_s)) + 1
            feature_workers = max(1, min(want, (os.cpu_count() or 4) - 2, 16))
        self.feature_workers = int(feature_workers)
        self.features = export.FeatureStore(self.out_dir / "features.h5")
        self._fstores = [self.features] + ([] if self.features.use_h5 else
                                           [export.FeatureStore(self.out_dir / "features.h5", shard=i) for i in range(1, self.feature_workers)])
        self.raw = export.MatchStore(self.out_dir / "raw_matches.h5")
        self.verified = export.MatchStore(self.out_dir / "matches.h5")
        self._write_db, self._camera_model = write_database, camera_model
        self._ids = {n: i + 1 for i, n in enumerate(sorted(image_names))} if image_names is not None else None
        self._db: Optional[export.ColmapDatabase] = None
        self._db_pairs = (set(), set())
        self._kpts: Dict[str, np.ndarray] = {}
        self._wh: Dict[str, Tuple[int, int]] = {}
        self._raw: Dict[Tuple[str, str], np.ndarray] = {}
        self._ver: Dict[Tuple[str, str], np.ndarray] = {}
        self._q: "queue.Queue" = queue.Queue()      # match batches + image rows -> ONE writer (ordered appends, sqlite connection)
        self._fq: "queue.Queue" = queue.Queue()     # per-image feature jobs -> the deflate pool
        self._lock = threading.Lock()
        self._h5_lock = threading.Lock()
        self._err: Optional[BaseException] = None
        cuda = self.device.type == "cuda"
        self._copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        self._fring, self._mring = _Ring(max_pending, cuda), _Ring(max_pending, cuda)
        self._dev_pack: List[Optional[torch.Tensor]] = [None, None]      # device-side packed slots, double-buffered
        self._dev_ver: List[Optional[tuple]] = [None, None]
        self._dev_events: List[List[Optional["torch.cuda.Event"]]] = [[None, None], [None, None]]
        self._turn = [0, 0]
        self.busy_s = 0.0            # summed over the writer threads (hidden behind the GPU when they keep up)
        self.n_images = self.n_pairs = self.n_verified_pairs = 0
        self._threads = [threading.Thread(target=self._run_matches, name="dim-writer-matches", daemon=True)]
        self._threads += [threading.Thread(target=self._run_features, args=(self._fstores[i % len(self._fstores)],), name=f"dim-writer-features{i}",
                                           daemon=True) for i in range(self.feature_workers)]
        for th in self._threads:
            th.start()

    # ---- producer side (GPU thread) --------------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None

    def _check(self):
        if self._err is not None:
            raise RuntimeError("a writer thread failed") from self._err

    def _dev_slot(self, which: int, alloc):
        """Double-buffered device scratch: returns (index, tensors) after the copy that last read it has been ordered before
        the kernels about to overwrite it (stream-side wait: the host does not block)."""
        i = self._turn[which]
        self._turn[which] ^= 1
        ev = self._dev_events[which][i]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return i, alloc(i)

    def _d2h(self, which: int, idx: int, pairs: Sequence[Tuple[torch.Tensor, torch.Tensor]]):
        """Asynchronous copies device -> ring buffers on the copy stream; returns the event that completes them."""
        if self._copy_stream is None:
            for src, dst in pairs:
                dst[: src.numel()].copy_(src.reshape(-1))
            return None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            for src, dst in pairs:
                dst[: src.numel()].copy_(src.reshape(-1), non_blocking=True)
                src.record_stream(self._copy_stream)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._dev_events[which][idx] = done
        return done

    @torch.no_grad()
    def put_features(self, names: Sequence[str], kpts, scores, desc, n, image_hw: Sequence[Tuple[int, int]], tile_idx=None):
        """One extracted batch: kpts [B,cap,2], scores [B,cap], desc [B,cap,D], n [B] int32 (device) -> features.h5 groups."""
        self._check()
        B, cap, D = int(desc.shape[0]), int(desc.shape[1]), int(desc.shape[2])
        slot = int(self.lib.dim_pack_features_slot_halves(cap, D))

        def alloc(i):
            t = self._dev_pack[i]
            if t is None or t.numel() < B * slot or t.device != kpts.device:
                t = self._dev_pack[i] = torch.empty(B * slot, dtype=torch.float16, device=kpts.device)
            return t

        if self.device.type == "cuda":
            idx, packed = self._dev_slot(0, alloc)
        else:
            idx, packed = 0, torch.empty(B * slot, dtype=torch.float16)
        ti = tile_idx.to(torch.int32).contiguous() if tile_idx is not None else None
        ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
        with ctx:
            capi.check(self.lib, self.lib.dim_op_pack_features_f16(capi.ptr(kpts.contiguous()), capi.ptr(scores.contiguous()), capi.ptr(desc.contiguous()),
                                                                   capi.ptr(n.contiguous()), capi.ptr(ti), B, cap, D, capi.ptr(packed), self._stream()))
        host = self._fring.acquire([((B * slot,), torch.float16), ((B,), torch.int32)], self._check)
        ev = self._d2h(0, idx, [(packed[: B * slot], host[0]), (n.to(torch.int32), host[1])])
        left = [len(names)]
        for b, name in enumerate(names):
            self._fq.put((ev, host, left, b, name, tuple(image_hw[b]), cap, D, slot))

    @torch.no_grad()
    def put_matches(self, pair_names: Sequence[Tuple[str, str]], matches, n_matches, mask=None):
        """One matched (and optionally verified) batch: matches [P,NK,2] int64, n_matches [P] int32, mask [P,NK] uint8 or None.
        With a mask the accept / reject rules of matcher_base.py:287-334 are applied on the device."""
        self._check()
        P, NK = int(matches.shape[0]), int(matches.shape[1])
        srcs = [matches, n_matches]
        idx = 0
        if mask is not None:
            def alloc(i):
                t = self._dev_ver[i]
                if t is None or t[0].shape != matches.shape or t[0].device != matches.device:
                    t = self._dev_ver[i] = (torch.empty_like(matches), torch.empty(P, dtype=torch.int32, device=matches.device))
                return t

            idx, (ver, n_ver) = self._dev_slot(1, alloc) if self.device.type == "cuda" else (0, alloc(0))
            ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
            with ctx:
                capi.check(self.lib, self.lib.dim_op_filter_matches(capi.ptr(matches), capi.ptr(n_matches), capi.ptr(mask.contiguous()), NK, P,
                                                                    self.min_inliers, ctypes.c_double(self.min_ratio), capi.ptr(ver), capi.ptr(n_ver),
                                                                    self._stream()))
            srcs += [ver, n_ver]
        host = self._mring.acquire([((t.numel(),), t.dtype) for t in srcs], self._check)
        ev = self._d2h(1, idx, list(zip(srcs, host)))
        self._q.put(("matches", ev, host, list(pair_names), P, NK, mask is not None))

    # ---- writer threads -------------------------------------------------------------------------------------------
    def _run_features(self, store):
        while True:
            item = self._fq.get()
            if item is None:
                self._fq.task_done()
                return
            ev, host, left, b, name, hw, cap, D, slot = item
            try:
                t0 = time.perf_counter()
                if ev is not None:
                    ev.synchronize()
                if self._err is None:     # after a failure the remaining jobs only hand their buffers back
                    self._write_features(store, host, b, name, hw, cap, D, slot)
                with self._lock:
                    self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001 - surfaced on the producer side (_check / close)
                self._err = self._err or e
            finally:
                with self._lock:          # the batch's pinned buffer goes back to the ring whether or not the write succeeded
                    left[0] -= 1
                    done = left[0] == 0
                if done:
                    self._fring.release(host)
                self._fq.task_done()

    def _write_features(self, store, host, b, name, hw, cap, D, slot):
        k = min(int(host[1][b]), cap)
        s = host[0].numpy()[b * slot:(b + 1) * slot]
        half = {"keypoints": s[: 2 * k].reshape(k, 2), "descriptors": s[4 * cap: 4 * cap + D * k].reshape(D, k), "scores": s[2 * cap: 2 * cap + k],
                "tile_idx": s[3 * cap: 3 * cap + k], "image_size": np.array(hw).astype(np.float16)}     # (H, W), extractor_base.py:227 (Q4)
        if store.use_h5:
            blobs = {key: zlib.compress(np.ascontiguousarray(v).tobytes(), 9) for key, v in half.items()}     # in parallel, GIL released
            with self._h5_lock:
                store.add_precompressed(name, half, blobs)
        else:
            store.add_half(name, half)
        kp32 = half["keypoints"].astype(np.float32)    # io/h5_to_db.py reads the fp16 keypoints back and stores them as float32
        wh = (int(hw[1]), int(hw[0]))
        with self._lock:
            self.n_images += 1
            if self._ids is None:
                self._kpts[name], self._wh[name] = kp32, wh
        if self._ids is not None and self._write_db:
            self._q.put(("image", name, kp32, wh))

    def _run_matches(self):
        while True:
            item = self._q.get()
            if item is None:
                try:
                    if self._db is not None:     # commit + close in the thread that owns the sqlite connection
                        self._db.close()
                except BaseException as e:  # noqa: BLE001
                    self._err = e
                self._q.task_done()
                return
            try:
                t0 = time.perf_counter()
                if item[0] == "image":
                    self._db_image(*item[1:])
                else:
                    _, ev, host, pair_names, P, NK, verified = item
                    try:
                        if ev is not None:
                            ev.synchronize()
                        if self._err is None:
                            self._write_matches(host, pair_names, P, NK, verified)
                    finally:
                        self._mring.release(host)
                with self._lock:
                    self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001
                self._err = self._err or e
            finally:
                self._q.task_done()

    def _database(self) -> export.ColmapDatabase:
        if self._db is None:   # created in the writer thread: sqlite connections belong to the thread that opened them
            self._db = export.ColmapDatabase(self.out_dir / "database.db")
        return self._db

    def _db_image(self, name, kp32, wh):
        db, iid = self._database(), self._ids[name]
        cam = db.add_default_camera(self._camera_model, wh[0], wh[1], camera_id=iid)
        db.add_image(name, cam, image_id=iid)
        if kp32.ndim >= 2:
            db.add_keypoints(iid, kp32)

    def _write_matches(self, host, pair_names, P, NK, verified):
        m = host[0].numpy()[: P * NK * 2].reshape(P, NK, 2)
        n = host[1].numpy()
        if verified:
            v, nv = host[2].numpy()[: P * NK * 2].reshape(P, NK, 2), host[3].numpy()
        live_db = self._ids is not None and self._write_db
        for p, (a, b) in enumerate(pair_names):
            s = min(int(n[p]), NK)
            raw = m[p, :s].copy()
            self.raw.add(a, b, raw)
            ver = v[p, : int(nv[p])].copy() if verified and int(nv[p]) >= 0 else None     # -1: dropped by the reference's rules
            if ver is not None:
                self.verified.add(a, b, ver)
                self.n_verified_pairs += 1
            if live_db:
                db = self._database()
                pid = export.image_ids_to_pair_id(self._ids[a], self._ids[b])
                if pid not in self._db_pairs[0]:        # duplicates are skipped like io/h5_to_db.py:286-292
                    db.add_matches(self._ids[a], self._ids[b], raw)
                    self._db_pairs[0].add(pid)
                if ver is not None and pid not in self._db_pairs[1]:
                    db.add_two_view_geometry(self._ids[a], self._ids[b], ver)
                    self._db_pairs[1].add(pid)
            elif self._write_db:
                self._raw[(a, b)] = raw
                if ver is not None:
                    self._ver[(a, b)] = ver
            self.n_pairs += 1

    # ---- shutdown ------------------------------------------------------------------------------------------------
    def close(self) -> Dict[str, float]:
        """Drains the queues, finalises the containers and the database; returns the writers' statistics."""
        self._fq.join()               # feature workers may still enqueue image rows for the database ...
        for _ in range(self.feature_workers):
            self._fq.put(None)
        self._q.join()                # ... which the match writer drains here
        t0 = time.perf_counter()
        self._q.put(None)             # the match writer commits and closes its database connection on the way out
        for th in self._threads:
            th.join()
        if self._err is not None:     # close the containers first (what was written stays readable), then report
            for st in self._fstores + [self.raw, self.verified]:
                try:
                    st.close()
                except BaseException:  # noqa: BLE001 - the first error is the one reported
                    pass
            raise RuntimeError("a writer thread failed") from self._err
        for st in self._fstores:
            st.close()
        self.raw.close(); self.verified.close()
        if self._write_db and self._ids is None and self._kpts:
            names = sorted(self._kpts)
            export.export_to_colmap(self.out_dir / "database.db", names, self._wh, self._kpts, self._raw, self._ver or None,
                                    camera_model=self._camera_model)
        return {"writer_busy_s": self.busy_s, "finalise_s": time.perf_counter() - t0, "images": self.n_images, "pairs": self.n_pairs,
                "verified_pairs": self.n_verified_pairs, "feature_workers": self.feature_workers, "deflate_mb_per_s": round(self.deflate_mb_per_s, 1)}


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class EndToEndRunner:
    """extract -> (all-gather) -> match -> verify -> export for one rank, with the writers and the verification off the
    critical path.  ``extractor`` / ``matcher``: SuperPointHIP / LightGlueHIP; ``verifier``: verify.DeviceVerifier.
    Every batch runs under the fp16x3 range guard (capi.run_guarded: a batch that leaves the exact range of the split is
    repeated in bf16x6 BEFORE it is handed to the exporter); the number of repeated batches is reported."""

    def __init__(self, extractor, matcher, verifier=None, exporter: Optional[AsyncExporter] = None, on_saturation: str = "fallback",
                 overlap_verification: bool = True):
        self.ext, self.mat, self.ver, self.exp = extractor, matcher, verifier, exporter
        self.policy = on_saturation
        self.overlap_verification = overlap_verification

    @torch.no_grad()
    def run(self, names: Sequence[str], images: torch.Tensor, pairs: torch.Tensor) -> Dict[str, float]:
        """images [n,H,W] float32 in [0,1] on the device; pairs [P,2] int32 image indices.  Returns timings (seconds)."""
        dev = images.device
        lib = self.ext.lib
        n_img, H, W = images.shape
        cap, B, PB = self.ext.capacity, self.ext.max_batch, self.mat.max_pairs
        sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
        stream = (lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) if dev.type == "cuda" else (lambda: None)
        reruns = [0]

        class _Count:
            def warning(self, *a):
                reruns[0] += 1

        sync()
        t0 = time.perf_counter()
        kp = torch.zeros(n_img, cap, 2, device=dev); sc = torch.zeros(n_img, cap, device=dev)
        de = torch.zeros(n_img, cap, 256, device=dev); n = torch.zeros(n_img, dtype=torch.int32, device=dev)
        for s in range(0, n_img, B):
            e = min(n_img, s + B)
            chunk = images[s:e].contiguous()
            k_, s_, d_, n_ = capi.run_guarded(lib, stream(), lambda: self.ext.extract_batch(chunk), "EndToEndRunner.extract", self.policy, _Count(),
                                                  handle=self.ext._h, arithmetic=getattr(self.ext, "arithmetic", None))
            kp[s:e], sc[s:e], de[s:e], n[s:e] = k_, s_, d_, n_
            if self.exp is not None:
                self.exp.put_features(names[s:e], kp[s:e], sc[s:e], de[s:e], n[s:e], [(H, W)] * (e - s))
        size = torch.tensor([[float(H), float(W)]] * n_img, device=dev)
        sync()
        t1 = time.perf_counter()
        pairs_dev = pairs.to(dev, torch.int32).contiguous()
        pair_list = pairs.tolist()
        counts = []
        # verification + hand-over to the writers run on their OWN stream: the fp64-VALU RANSAC of batch i overlaps the matrix-core
        # bound LightGlue of batch i + 1 (measured in round 4 with ~1400 matches per pair: RANSAC on the matching stream cost 14 %)
        vstream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.ver is not None and self.overlap_verification) else None
        for s in range(0, pairs.shape[0], PB):
            pp = pairs_dev[s:s + PB].contiguous()
            o = capi.run_guarded(lib, stream(), lambda: self.mat.match_batch(kp, de, n, size, pair_idx=pp), "EndToEndRunner.match", self.policy, _Count(),
                                 handle=self.mat._h, arithmetic=getattr(self.mat, "arithmetic", None))
            if vstream is not None:
                vstream.wait_stream(torch.cuda.current_stream(dev))
                for t_ in (o["matches"], o["n_matches"], pp):
                    t_.record_stream(vstream)
            with (torch.cuda.stream(vstream) if vstream is not None else _Null()):
                mask = None
                if self.ver is not None:
                    v = self.ver.verify_batch(kp, o["matches"], o["n_matches"], pair_idx=pp)
                    mask = v["mask"]
                    counts.append((o["n_matches"].clone(), v["n_inliers"].clone()))
                else:
                    counts.append((o["n_matches"].clone(), None))
                if self.exp is not None:
                    pn = [(names[a], names[b]) for a, b in pair_list[s:s + PB]]
                    self.exp.put_matches(pn, o["matches"], o["n_matches"], mask)
        if vstream is not None:
            torch.cuda.current_stream(dev).wait_stream(vstream)
        sync()
        t2 = time.perf_counter()
        stats = self.exp.close() if self.exp is not None else {}
        t3 = time.perf_counter()
        tot_raw = sum(int(a.sum().item()) for a, _ in counts)
        tot_ver = sum(int(b.sum().item()) for _, b in counts if b is not None)
        P = int(pairs.shape[0])
        return {"images": n_img, "pairs": P, "extract_s": t1 - t0, "match_verify_s": t2 - t1, "drain_s": t3 - t2,
                "kernel_path_pairs_per_s": P / (t2 - t1), "end_to_end_pairs_per_s": P / (t3 - t0), "raw_matches": tot_raw,
                "verified_matches": tot_ver, "guard_reruns": reruns[0], **stats}
