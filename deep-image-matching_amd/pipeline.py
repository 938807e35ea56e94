"""Data-parallel extraction + matching over the GPUs of one node.

The reference is single-process / single-device (image_matching.py:413-494: one
``extract`` per image, one ``match`` per pair).  Here one process drives one GPU and

  phase 1  images  i ≡ rank (mod world) are extracted locally (batched dim_sp_extract),
  phase 2  ONE all-gather makes every rank hold all features: the extractor writes its outputs straight into the
           sections [kpts | scores | descriptors | counts] of one flat fp32 buffer per rank (fixed slots,
           n_img/world x cap x (2 + 1 + D) floats + the counts' bit patterns), which is exchanged as it is,
  phase 3  the pair list (``itertools.combinations`` order for bruteforce,
           pairs_generator.py:37-38) is sharded round-robin, each rank matches its shard
           in batches (dim_lg_match with a pair-index table: no feature copies),
  phase 4  ONE all-gather of the per-rank match tables gives every rank the complete result: a flat int32
           buffer [counts | stop | (idx0, idx1, score bits) rows] (12-byte rows, zero beyond a
           pair's count; packed and un-packed by dim_op_pack_match_rows / dim_op_unpack_match_rows, the latter
           also undoing the round-robin shard order).

No collective sits on the per-pair data path.  ``torch.distributed`` backend "nccl" is
RCCL over xGMI on the GPU box; the same code runs under "gloo" on CPU tensors in the
world_size-2 tests (with the emulator-built library injected).
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Sequence, Tuple

import torch


def exhaustive_pairs(n_images: int, limit: Optional[int] = None) -> torch.Tensor:
    """Bruteforce pair list in the reference's order (pairs_generator.py:37-38)."""
    it = itertools.combinations(range(n_images), 2)
    if limit is not None:
        it = itertools.islice(it, limit)
    p = torch.tensor(list(it), dtype=torch.int32)
    return p.reshape(-1, 2)


def shard_indices(n_items: int, rank: int, world: int) -> torch.Tensor:
    """Round-robin shard (item i -> rank i % world): equal counts +-1, order preserved."""
    return torch.arange(rank, n_items, world, dtype=torch.long)


def balanced_shards(costs: torch.Tensor, world: int):
    """Cost-balanced shards with equal counts (+-1): items sorted by cost (descending, stable) are dealt to the ranks in serpentine order
    (0 .. w-1, w-1 .. 0, ...), each rank keeping its items in ascending index order.  Returns (rank_of [n], slot_of [n]) on the CPU.
    SURVEY 8(e): LightGlue's work per pair scales with n0 x n1 (and, with adaptive depth, with the data); round-robin by index can leave one
    rank with all the heavy pairs.  Equal costs reproduce the plain round-robin (item i -> rank i mod world, slot i div world)."""
    n = int(costs.numel())
    order = torch.argsort(costs.to(torch.float64).cpu(), descending=True, stable=True)
    k = torch.arange(n)
    lap, pos = k // world, k % world
    deal = torch.where(lap % 2 == 0, pos, world - 1 - pos)
    if bool((costs.reshape(-1)[:1].expand(n).cpu() == costs.reshape(-1).cpu()).all()):     # all equal: keep the documented round-robin
        deal, order = k % world, k
    rank_of = torch.empty(n, dtype=torch.long)
    rank_of[order] = deal
    slot_of = torch.empty(n, dtype=torch.long)
    for r in range(world):
        idx = torch.nonzero(rank_of == r).reshape(-1)        # ascending index order inside a rank
        slot_of[idx] = torch.arange(idx.numel())
    return rank_of, slot_of


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def _all_gather_cat(t: torch.Tensor, world: int) -> torch.Tensor:
    """all_gather of equally-shaped tensors, concatenated along dim 0 (rank-major)."""
    dist = _dist()
    if dist is None or world == 1:
        return t
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def _guarded(net, fn, what: str):
    """fn() under the fp16x3 range guard of ``net`` (capi.run_guarded): one counter read-back per phase; a phase that
    left the exact range of the fp16 split is repeated in bf16x6."""
    from . import capi

    with net._ctx():
        return capi.run_guarded(net.lib, net._stream(), fn, what, getattr(net, "on_saturation", "fallback"))


class PairMatchingPipeline:
    """extractor: SuperPointHIP, matcher: LightGlueHIP (both resident on this rank's device)."""

    def __init__(self, extractor, matcher, rank: int = 0, world: int = 1):
        self.ext, self.mat, self.rank, self.world = extractor, matcher, rank, world
        self.timings: dict = {}     # per-phase wall times of the last extract_all / match_all on this rank (seconds) + gathered bytes

    # ---- phases 1+2 ------------------------------------------------------------------------
    @torch.no_grad()
    def extract_all(self, images: torch.Tensor, image_sizes: Optional[torch.Tensor] = None):
        """images [n_img, H, W] float32 in [0,1], identical on every rank (or at least the
        rank's own shard valid).  Returns the GLOBAL feature table (kpts [n_img,cap,2],
        scores [n_img,cap], desc [n_img,cap,D], n [n_img], size [n_img,2]) on every rank."""
        import time
        n_img, H, W = images.shape
        mine = shard_indices(n_img, self.rank, self.world)
        per = (n_img + self.world - 1) // self.world
        cap, dev, D = self.ext.capacity, images.device, 256
        # one flat buffer per rank, sections [kp | sc | de | n]: the extractor writes into views of it, the collective ships it whole
        sizes = (per * cap * 2, per * cap, per * cap * D, per)
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        o = [0]
        for z in sizes:
            o.append(o[-1] + z)
        kp, sc, de = flat[o[0]:o[1]].view(per, cap, 2), flat[o[1]:o[2]].view(per, cap), flat[o[2]:o[3]].view(per, cap, D)
        n = flat[o[3]:o[4]].view(torch.int32)
        B = self.ext.max_batch
        t0 = time.perf_counter()

        def run():  # every batch of the shard is enqueued back to back; the fp16x3 range guard is read once per phase
            for s in range(0, len(mine), B):
                idx = mine[s:s + B]
                b = len(idx)
                self.ext.extract_batch(images[idx.to(dev)].contiguous(), out=(kp[s:s + b], sc[s:s + b], de[s:s + b], n[s:s + b]))

        _guarded(self.ext, run, "pipeline extraction")     # synchronises (guard read-back)
        t1 = time.perf_counter()
        g = _all_gather_cat(flat[None], self.world)         # phase 2: ONE collective, [world, flat]
        if self.world > 1:
            # slot (r, j) holds image j*world + r: gather the sections into image order
            order = torch.arange(self.world * per, device=dev).reshape(self.world, per).t().reshape(-1)[:n_img]
            kp_g = g[:, o[0]:o[1]].reshape(self.world * per, cap, 2)[order]
            sc_g = g[:, o[1]:o[2]].reshape(self.world * per, cap)[order]
            de_g = g[:, o[2]:o[3]].reshape(self.world * per, cap, D)[order]
            n_g = g[:, o[3]:o[4]].reshape(self.world * per).view(torch.int32)[order]
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
        else:
            kp_g, sc_g, de_g, n_g = kp[:n_img], sc[:n_img], de[:n_img], n[:n_img]
        self.timings.update(extract_s=t1 - t0, feature_gather_s=time.perf_counter() - t1, feature_gather_bytes=int(flat.numel() * 4 * self.world))
        if image_sizes is None:  # DIM stores image.shape[:2] = (H, W) (extractor_base.py:227, Q4)
            image_sizes = torch.tensor([[float(H), float(W)]] * n_img)
        return kp_g.contiguous(), sc_g.contiguous(), de_g.contiguous(), n_g.contiguous(), image_sizes.to(dev, torch.float32).contiguous()

    # ---- phases 3+4 ------------------------------------------------------------------------
    @torch.no_grad()
    def match_all(self, table, pairs: torch.Tensor, aux: bool = False):
        """table from extract_all (or any device feature table); pairs [P,2] int32 (image slots).
        Returns, on every rank, (n_matches [P], matches [P,NK,2] int64, scores [P,NK]) in the
        order of ``pairs``; with ``aux`` additionally (stop [P] int32, prune01 [P,2,NK] int32) — the reference's
        "stop" / "prune0" / "prune1" outputs (LGN:570-577), gathered the same way (parity tests)."""
        import ctypes
        import time
        from . import capi
        kp, sc, de, n, size = table
        dev = kp.device
        lib = self.mat.lib
        P = pairs.shape[0]
        # pairs dealt to the ranks by cost n0 x n1 (equal counts; equal costs = round-robin): VERDICT r3 weak #13
        pl_ = pairs.to(torch.long).cpu()
        n_h = n.cpu().to(torch.float64)
        rank_of, slot_of = balanced_shards(n_h[pl_[:, 0]] * n_h[pl_[:, 1]], self.world) if P else (torch.zeros(0, dtype=torch.long),) * 2
        mine = torch.nonzero(rank_of == self.rank).reshape(-1)
        per = (P + self.world - 1) // self.world
        NK, B = self.mat.nk, self.mat.max_pairs
        # flat int32 buffer per rank: [cnt: per | stop: per | rows: per*NK*3 | (aux) prune: per*2*NK]
        sizes = (per, per, per * NK * 3, per * 2 * NK if aux else 0)
        flat = torch.zeros(sum(sizes), dtype=torch.int32, device=dev)
        o = [0]
        for z in sizes:
            o.append(o[-1] + z)
        cnt, stp, rows = flat[o[0]:o[1]], flat[o[1]:o[2]], flat[o[2]:o[3]].view(per, NK, 3)
        prn = flat[o[3]:o[4]].view(per, 2, NK) if aux else None
        my_pairs = pairs[mine].to(dev, torch.int32).contiguous()
        stream = (lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) if dev.type == "cuda" else (lambda: None)
        t0 = time.perf_counter()

        def run():
            out = None
            for s in range(0, len(mine), B):
                pp = my_pairs[s:s + B].contiguous()
                b = pp.shape[0]
                out = self.mat.match_batch(kp, de, n, size, pair_idx=pp, n_pairs=b, out=out)   # the first chunk is the largest: buffers are reused
                with self.mat._ctx():
                    capi.check(lib, lib.dim_op_pack_match_rows(capi.ptr(out["matches"]), capi.ptr(out["scores"]), capi.ptr(out["n_matches"]), NK, b,
                                                                capi.ptr(rows[s:s + b]), stream()))
                cnt[s:s + b] = out["n_matches"][:b]
                stp[s:s + b] = out["stop"][:b]
                if aux:
                    prn[s:s + b] = out["prune01"][:b]

        _guarded(self.mat, run, "pipeline matching")       # synchronises
        t1 = time.perf_counter()
        g = _all_gather_cat(flat[None], self.world)         # phase 4: ONE collective, [world, flat]
        # pair p was matched by rank rank_of[p] as its slot_of[p]-th pair
        src = (rank_of * per + slot_of).to(dev)
        cnt_g = g[:, o[0]:o[1]].reshape(-1)[src].contiguous()
        stp_g = g[:, o[1]:o[2]].reshape(-1)[src].contiguous()
        rows_g = g[:, o[2]:o[3]].reshape(self.world * per, NK, 3) if self.world > 1 else rows
        mt_g = torch.empty(P, NK, 2, dtype=torch.int64, device=dev)
        ms_g = torch.empty(P, NK, dtype=torch.float32, device=dev)
        rows_c = rows_g.contiguous()
        src32 = src.to(torch.int32).contiguous()
        with self.mat._ctx():
            capi.check(lib, lib.dim_op_unpack_match_rows(capi.ptr(rows_c), capi.ptr(src32), NK, P, capi.ptr(mt_g), capi.ptr(ms_g), stream()))
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        self.timings.update(match_s=t1 - t0, match_gather_s=time.perf_counter() - t1, match_gather_bytes=int(flat.numel() * 4 * self.world))
        if aux:
            prn_g = g[:, o[3]:o[4]].reshape(self.world * per, 2, NK)[src].contiguous()
            return cnt_g, mt_g, ms_g, stp_g, prn_g
        return cnt_g, mt_g, ms_g

    @staticmethod
    def to_match_lists(cnt: torch.Tensor, mt: torch.Tensor, ms: torch.Tensor) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """Host-side unpadding: [(matches (S,2) int64, scores (S,)) per pair] — the arrays DIM
        writes to raw_matches.h5 (matcher_base.py:282-285)."""
        cnt = cnt.cpu()
        mt, ms = mt.cpu(), ms.cpu()
        return [(mt[p, : int(cnt[p])], ms[p, : int(cnt[p])]) for p in range(cnt.shape[0])]


class TiledPairPipeline:
    """BASELINE config 5 across the GPUs of one node: tile-wise extraction + tile-pair matching of large images, sharded like
    PairMatchingPipeline with the TILE TABLE of an image as the feature unit.

      phase 1  images i = rank (mod world): the extractor plugin's batched ``_extract_by_tile`` (extractors/extractor_base.py:279-390:
               pad, unfold, one forward per tile, shift, border filter, np.unique merge — here one batch per image on the device),
      phase 2  ONE all-gather of a flat fp32 buffer per rank: [per][cap][keypoints 2 | score | tile_idx | descriptor D] + the counts'
               bit patterns (cap = tiles x max keypoints per tile: the merged table of an image can hold no more),
      phase 3  image pairs j = rank (mod world): ``tile_selection`` (matchers/matcher_base.py:989-1140; PRESELECTION runs its own
               down-sampled SuperPoint + LightGlue on the device) and the batched tile-pair matching (MB:362-485),
      phase 4  ONE all-gather of a flat int32 buffer per rank: [per][count | (idx0, idx1) rows].

    ``extractor`` / ``matcher``: plugins.SuperPointExtractor / AlikedExtractor and plugins.LightGlueMatcher (any descriptor width and
    channel count: both come from the extractor).  Every rank needs the image arrays of the pairs it matches only when the selection
    method reads pixels (PRESELECTION*).  Results are identical on every rank and identical to a single-process run."""

    def __init__(self, extractor, matcher, rank: int = 0, world: int = 1, selection: str = "PRESELECTION", max_kpts_per_image: Optional[int] = None,
                 max_matches_per_pair: Optional[int] = None, empty_selection_fallback: Optional[str] = None):
        self.ext, self.mat, self.rank, self.world = extractor, matcher, rank, world
        self.selection = selection
        # benchmarks on seeded synthetic weights only: PRESELECTION needs a trained SuperPoint + LightGlue to vote for tile pairs; with
        # random weights it runs (and is timed) but selects nothing, and the named method then supplies the tile pairs.  None (default,
        # the reference's behaviour): an empty selection means an empty match list.
        self.fallback = empty_selection_fallback
        self.n_fallback = 0
        self.max_kpts, self.max_matches = max_kpts_per_image, max_matches_per_pair
        self.timings: dict = {}

    def _band(self, images, i):
        """first band of image i as a contiguous float32 array, cached (extracting 1 of 3 interleaved channels of a 6000 x 4000 image
        costs ~40 ms; every image takes part in n - 1 pairs)"""
        c = self.__dict__.setdefault("_band_cache", {})
        key = (id(images), i)
        if key not in c:
            c[key] = _band1(images[i])
        return c[key]

    def _device(self):
        d = getattr(self.ext, "_device", "cuda")
        return torch.device(d if isinstance(d, (str, torch.device)) else "cuda")

    def _cap(self, image) -> int:
        if self.max_kpts is not None:
            return int(self.max_kpts)
        from .tile_matching import tile_grid
        general = self.ext.config["general"]
        n_tiles = len(tile_grid(image.shape[:2], general["tile_size"], general.get("tile_overlap", 0)))
        mk = int(self.ext.config["extractor"].get("max_num_keypoints", self.ext.config["extractor"].get("max_keypoints", -1)))
        if mk <= 0:
            raise ValueError("TiledPairPipeline: pass max_kpts_per_image when the extractor keeps all keypoints")
        return n_tiles * mk

    # ---- phases 1 + 2 ----------------------------------------------------------------------
    @torch.no_grad()
    def extract_all(self, images: Sequence, as_numpy: bool = True) -> List[dict]:
        """images: sequence of numpy arrays (H, W) or (H, W, C), 0..255, the same list on every rank (only this rank's shard is read).
        Returns the feature dict of EVERY image (keypoints (N,2) f32, descriptors (D,N) f32, scores (N,), tile_idx (N,), image_size) as
        numpy arrays; with ``as_numpy=False`` the device views of the exchange buffer that match_all uses anyway (keypoints [N,2],
        descriptors_nd [N,D], tile_idx, scores: no device-to-host copy of 34 MB per image)."""
        import time
        import numpy as np
        n_img = len(images)
        mine = shard_indices(n_img, self.rank, self.world).tolist()
        per = (n_img + self.world - 1) // self.world
        D, dev = int(self.ext.descriptor_size), self._device()
        cap = self._cap(images[0])
        row = 2 + 1 + 1 + D
        flat = torch.zeros(per * cap * row + per, dtype=torch.float32, device=dev)
        body, cnt = flat[: per * cap * row].view(per, cap, row), flat[per * cap * row:].view(torch.int32)
        t0 = time.perf_counter()
        for s, i in enumerate(mine):
            # the merged tile table never leaves HBM: merge_tile_features_device -> views of the exchange buffer
            f = self.ext._extract_by_tile(np.asarray(images[i]), as_device=True)
            k = int(f["keypoints"].shape[0])
            if k > cap:
                raise ValueError(f"TiledPairPipeline: image {i} has {k} keypoints, more than the exchange slot ({cap})")
            body[s, :k, 0:2] = f["keypoints"]
            body[s, :k, 2] = f["scores"]
            body[s, :k, 3] = f["tile_idx"]
            body[s, :k, 4:] = f["descriptors"].t()
            cnt[s] = k
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        g = _all_gather_cat(flat[None], self.world)          # phase 2: ONE collective
        out: List[dict] = []
        gb = g[:, : per * cap * row].reshape(self.world, per, cap, row)
        gc = (g[:, per * cap * row:].reshape(self.world, per).view(torch.int32) if self.world > 1 else cnt.view(1, per)).cpu()
        self._dev_feats = []
        for i in range(n_img):
            r, s = i % self.world, i // self.world
            k = int(gc[r, s])
            size = np.array(np.asarray(images[i]).shape[:2], dtype=np.int32)
            t = gb[r, s, :k]
            self._dev_feats.append({"keypoints": t[:, 0:2], "descriptors_nd": t[:, 4:], "tile_idx": t[:, 3], "scores": t[:, 2], "image_size": size})
            if as_numpy:
                h = t.cpu().numpy()
                out.append({"keypoints": np.ascontiguousarray(h[:, 0:2]), "scores": np.ascontiguousarray(h[:, 2]), "tile_idx": np.ascontiguousarray(h[:, 3]),
                            "descriptors": np.ascontiguousarray(h[:, 4:].T), "image_size": size})
            else:
                out.append(self._dev_feats[-1])
        self.timings.update(extract_s=t1 - t0, feature_gather_s=time.perf_counter() - t1, feature_gather_bytes=int(flat.numel() * 4 * self.world))
        return out

    # ---- phases 3 + 4 ----------------------------------------------------------------------
    @torch.no_grad()
    def match_all(self, images: Sequence, feats: List[dict], pairs: torch.Tensor, names: Optional[Sequence[str]] = None) -> List:
        """pairs [P, 2] image indices.  Returns, on every rank, the list of (M, 2) int64 match arrays in the order of ``pairs``
        (the arrays MatcherBase._match_by_tile returns, MB:362-485)."""
        import time
        import numpy as np
        from .tile_matching import match_tile_pairs_batched, match_tile_pairs_batched_device
        P = int(pairs.shape[0])
        dev_feats = getattr(self, "_dev_feats", None)
        use_dev = dev_feats is not None and len(dev_feats) == len(feats)      # tables of the last extract_all are still in HBM
        mine = shard_indices(P, self.rank, self.world).tolist()
        per = (P + self.world - 1) // self.world
        dev = self._device()
        cap_m = int(self.max_matches if self.max_matches is not None else 2 * max(1, max(int(f["keypoints"].shape[0]) for f in feats)))
        flat = torch.zeros(per + per * cap_m * 2, dtype=torch.int32, device=dev)
        cnt, rows = flat[:per], flat[per:].view(per, cap_m, 2)
        names = names if names is not None else [f"image{i:05d}" for i in range(len(feats))]
        sel_s = mat_s = 0.0
        t0 = time.perf_counter()
        for s, p in enumerate(mine):
            a, b = int(pairs[p, 0]), int(pairs[p, 1])
            ts = time.perf_counter()
            needs_pixels = self.selection.startswith("PRESELECTION")
            shape_only = lambda i: np.broadcast_to(np.float32(0), np.asarray(images[i]).shape[:2])     # the grid methods read the shape only
            band = lambda i: self._band(images, i) if needs_pixels else shape_only(i)
            tile_pairs = self.mat.tile_selection(names[a], names[b], self.selection, image0=band(a), image1=band(b))
            if len(tile_pairs) == 0 and self.fallback is not None:
                self.n_fallback += 1
                tile_pairs = self.mat.tile_selection(names[a], names[b], self.fallback, image0=shape_only(a), image1=shape_only(b))
            tm_ = time.perf_counter()
            if use_dev:
                m = match_tile_pairs_batched_device(self.mat._ensure_pairs, dev_feats[a], dev_feats[b], tile_pairs, getattr(self.mat, "tile_pair_batch", 8))
            else:
                m = torch.from_numpy(match_tile_pairs_batched(self.mat._ensure_pairs, feats[a], feats[b], tile_pairs, dev, getattr(self.mat, "tile_pair_batch", 8))).to(dev)
            sel_s += tm_ - ts
            mat_s += time.perf_counter() - tm_
            if m.shape[0] > cap_m:
                raise ValueError(f"TiledPairPipeline: pair ({a}, {b}) has {m.shape[0]} matches, more than the exchange slot ({cap_m})")
            rows[s, : m.shape[0]] = m.to(torch.int32)
            cnt[s] = m.shape[0]
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        g = _all_gather_cat(flat[None], self.world)          # phase 4: ONE collective
        gc, gr = g[:, :per].cpu(), g[:, per:].reshape(self.world, per, cap_m, 2)
        out = []
        for p in range(P):
            r, s = p % self.world, p // self.world
            out.append(gr[r, s, : int(gc[r, s])].cpu().numpy().astype(np.int64))
        self.timings.update(match_s=t1 - t0, tile_selection_s=sel_s, tile_matching_s=mat_s, match_gather_s=time.perf_counter() - t1,
                            match_gather_bytes=int(flat.numel() * 4 * self.world))
        return out


def _band1(image):
    """the first band of an image array as float32 (what tile_selection reads with rasterio, MB:1021-1024)"""
    import numpy as np
    a = np.asarray(image)
    return np.ascontiguousarray(a if a.ndim == 2 else a[..., 0], dtype=np.float32)
