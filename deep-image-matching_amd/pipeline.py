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

import logging

import torch

logger = logging.getLogger("dim")


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
        # the handle and its own arithmetic override (plugin option `arithmetic`): the guard must decide from THAT handle's mode and re-run
        # THAT handle in bf16x6 — a handle override beats the process default inside the library
        return capi.run_guarded(net.lib, net._stream(), fn, what, getattr(net, "on_saturation", "fallback"), handle=getattr(net, "_h", None),
                                arithmetic=getattr(net, "arithmetic", None))


class PairMatchingPipeline:
    """extractor: SuperPointHIP (images [n, H, W]) or AlikedHIP (images [n, H, W, C]; its descriptor width is read from the extractor: 128, 64
    for aliked-t16), matcher: LightGlueHIP of the same input_dim (both resident on this rank's device)."""

    def __init__(self, extractor, matcher, rank: int = 0, world: int = 1):
        self.ext, self.mat, self.rank, self.world = extractor, matcher, rank, world
        self.timings: dict = {}     # per-phase wall times of the last extract_all / match_all on this rank (seconds) + gathered bytes

    # ---- phases 1+2 ------------------------------------------------------------------------
    @torch.no_grad()
    def extract_all(self, images: torch.Tensor, image_sizes: Optional[torch.Tensor] = None):
        """images [n_img, H, W] (SuperPoint) or [n_img, H, W, C] (ALIKED) float32 in [0,1], identical on every rank (or at least the
        rank's own shard valid).  Returns the GLOBAL feature table (kpts [n_img,cap,2],
        scores [n_img,cap], desc [n_img,cap,D], n [n_img], size [n_img,2]) on every rank."""
        import time
        n_img, H, W = images.shape[:3]
        mine = shard_indices(n_img, self.rank, self.world)
        per = (n_img + self.world - 1) // self.world
        cap, dev, D = self.ext.capacity, images.device, int(getattr(self.ext, "dim", 256))      # 256: SuperPoint; AlikedHIP.dim: 128 / 64
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
               bit patterns (cap = tiles x max keypoints per tile of the LARGEST image: the merged table of an image can hold no more),
      phase 3a ``tile_selection`` (matchers/matcher_base.py:989-1140) of image pairs j = rank (mod world); PRESELECTION runs its own
               down-sampled SuperPoint + LightGlue on the device for all of the rank's image pairs back to back (one read-back of the
               vote tables at the end), then ONE small all-gather of the selection masks [pairs][tiles0 x tiles1] bytes,
      phase 3b every rank deals the image pairs by cost — sum of n0 x n1 over the SELECTED tile pairs (balanced_shards) — and matches
               its image pairs' tile pairs as ONE stream of dim_lg_match batches over a per-image tile table that is built once per image
               (not once per image pair); tile-local match indices are mapped to merged-table indices, made unique per image pair and
               sorted like np.unique(axis=0) (MB:462-474) on the device — no host read-back per image pair or per batch,
      phase 4  ONE all-gather of a flat int32 buffer per rank: [per][count | (idx0, idx1) rows]; the slot size is the exact upper bound
               sum(min(n0, n1)) over the selected tile pairs, which every rank derives from the gathered masks (no heuristic, no abort).

    ``extractor`` / ``matcher``: plugins.SuperPointExtractor / AlikedExtractor and plugins.LightGlueMatcher (any descriptor width and
    channel count: both come from the extractor).  Every rank reads the SHAPES of all images; pixels only of its own extraction shard and —
    when the selection method reads pixels (PRESELECTION*) — of the image pairs whose selection it runs.  Results are identical on every
    rank and identical to a single-process run."""

    def __init__(self, extractor, matcher, rank: int = 0, world: int = 1, selection: str = "PRESELECTION", max_kpts_per_image: Optional[int] = None,
                 max_matches_per_pair: Optional[int] = None, empty_selection_fallback: Optional[str] = None, tile_pair_batch: Optional[int] = None):
        self.ext, self.mat, self.rank, self.world = extractor, matcher, rank, world
        self.selection = selection
        # the tile ids of the extracted features come from the FULL-resolution grid (extract_all does not resize), the selection grids from
        # get_size_by_quality(quality, shape): they are the same grid only at quality HIGH — anything else would select tiles that are
        # not the features' tiles
        for cfg in (extractor.config, matcher.config):
            q = cfg["general"].get("quality", "HIGH")
            if getattr(q, "name", q) != "HIGH":
                raise ValueError(f"TiledPairPipeline works on the full-resolution tiling (general.quality HIGH); got {getattr(q, 'name', q)!r} — "
                                 "use the plugins under the reference's own loops for the resized qualities")
        # benchmarks on seeded synthetic weights only: PRESELECTION needs a SuperPoint + LightGlue that can vote for tile pairs; when it
        # selects nothing, the named method supplies the tile pairs.  None (default, the reference's behaviour): an empty selection means
        # an empty match list.
        self.fallback = empty_selection_fallback
        self.n_fallback = 0
        self.max_kpts, self.max_matches = max_kpts_per_image, max_matches_per_pair
        self.tile_pair_batch = int(tile_pair_batch if tile_pair_batch is not None else getattr(matcher, "tile_pair_batch", 8))
        self.preselection_pair_batch = 16   # image pairs per LightGlue call of the PRESELECTION phase (down-sampled features, <= 4096 keypoints: ~0.1 GB of state per pair)
        self.timings: dict = {}
        self._dev_feats = None      # device views of the last extract_all's exchange buffer ...
        self._dev_token = None      # ... valid only for the list object that extract_all returned (no silent reuse)

    def _band(self, images, i):
        """first band of image i as a contiguous float32 array (extracting 1 of 3 interleaved channels of a 6000 x 4000 image costs ~40 ms;
        an image takes part in several pairs).  Keyed by the array object itself, dropped at the end of every match_all."""
        c = self.__dict__.setdefault("_band_cache", {})
        key = id(images[i])
        if key not in c or c[key][0] is not images[i]:
            c[key] = (images[i], _band1(images[i]))
        return c[key][1]

    def _device(self):
        d = getattr(self.ext, "_device", "cuda")
        return torch.device(d if isinstance(d, (str, torch.device)) else "cuda")

    def _n_tiles(self, shape_hw) -> int:
        from .tile_matching import tile_grid
        general = self.ext.config["general"]
        return len(tile_grid(tuple(shape_hw[:2]), general["tile_size"], general.get("tile_overlap", 0)))

    def _cap(self, images) -> int:
        """exchange slot = the largest merged tile table any image of the job can have"""
        if self.max_kpts is not None:
            return int(self.max_kpts)
        mk = int(self.ext.config["extractor"].get("max_num_keypoints", self.ext.config["extractor"].get("max_keypoints", -1)))
        if mk <= 0:
            raise ValueError("TiledPairPipeline: pass max_kpts_per_image when the extractor keeps all keypoints")
        return max(self._n_tiles(np_shape(im)) for im in images) * mk

    # ---- phases 1 + 2 ----------------------------------------------------------------------
    @torch.no_grad()
    def extract_all(self, images: Sequence, as_numpy: bool = True, names: Optional[Sequence[str]] = None) -> List[dict]:
        """images: sequence of numpy arrays (H, W) or (H, W, C), 0..255, the same list on every rank (pixels are read of this rank's shard
        only, shapes of all).  Returns the feature dict of EVERY image (keypoints (N,2) f32, descriptors (D,N) f32, scores (N,), tile_idx (N,),
        image_size) as numpy arrays; with ``as_numpy=False`` the device views of the exchange buffer (keypoints [N,2], descriptors_nd [N,D],
        tile_idx, scores: no device-to-host copy of 34 MB per image).  match_all uses the device tables when it is handed THIS list."""
        import time
        import numpy as np
        n_img = len(images)
        mine = shard_indices(n_img, self.rank, self.world).tolist()
        per = (n_img + self.world - 1) // self.world
        D, dev = int(self.ext.descriptor_size), self._device()
        cap = self._cap(images)
        row = 2 + 1 + 1 + D
        n_main = per * cap * row + per
        # PRESELECTION: the down-sampled SuperPoint features of an image (<= pcap keypoints: 4 MB at the reference's 4000) are derived from the device
        # copy the tiled extraction makes anyway, and TRAVEL WITH THE TILE TABLES: a third section of the exchange buffer, [per][kpts pcap x 2 | desc
        # pcap x 256] + counts + scales, +12 % of its size at config-5 shapes.  After the all-gather every rank files them in the preselector's cache under
        # the image's name (``names`` as in match_all; the same default), so the selection phase reads no pixels at all — extracting the first band of
        # a 6000 x 4000 x 3 float array costs ~20 ms on the host, which was 80 % of the selection phase of the config-5 benchmark and would be a third
        # of a rank's whole matching time at 8 ranks.
        pre = quality = keys = None
        pcap = psz = 0
        if self.selection.startswith("PRESELECTION") and hasattr(self.mat, "_preselector"):
            general = self.mat.config["general"]
            if general.get("preselection_pipeline", "superpoint+lightglue") == "superpoint+lightglue":
                quality = getattr(general.get("quality", "HIGH"), "name", general.get("quality", "HIGH"))
                pre = self.mat._preselector()
                keys = list(names) if names is not None else [f"image{i:05d}" for i in range(n_img)]
                pcap = pre._capacity()
                psz = pcap * (2 + 256)
                pre._cache_size = max(pre._cache_size, 2 * n_img)      # (entries of other ranks' images are views of the gathered buffer)
        flat = torch.zeros(n_main + (per * psz + 2 * per if pre is not None else 0), dtype=torch.float32, device=dev)
        body, cnt = flat[: per * cap * row].view(per, cap, row), flat[per * cap * row:n_main].view(torch.int32)
        if pre is not None:
            pbody = flat[n_main:n_main + per * psz].view(per, psz)
            pn, pscale = flat[n_main + per * psz:n_main + per * psz + per].view(torch.int32), flat[n_main + per * psz + per:]

        def pre_hook(i, s):
            def hook(src):
                pre._cache.pop((keys[i], quality), None)               # this call has the pixels: whatever an earlier job cached under the name is stale
                kp, de, n, scale = pre.features(keys[i], src if src.dim() == 2 else src[..., 0], quality)
                pbody[s, : pcap * 2] = kp.reshape(-1)
                pbody[s, pcap * 2:] = de.reshape(-1)
                pn[s] = n[0]
                pscale[s] = float(scale)
            return hook

        t0 = time.perf_counter()
        for s, i in enumerate(mine):
            # the merged tile table never leaves HBM: merge_tile_features_device -> views of the exchange buffer
            f = self.ext._extract_by_tile(np.asarray(images[i]), as_device=True, **({"on_device_image": pre_hook(i, s)} if pre is not None else {}))
            k = int(f["keypoints"].shape[0])
            if k > cap:
                raise ValueError(f"TiledPairPipeline: image {i} has {k} keypoints, more than the exchange slot ({cap}): max_kpts_per_image is too small")
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
        gc = (g[:, per * cap * row:n_main].reshape(self.world, per).view(torch.int32) if self.world > 1 else cnt.view(1, per)).cpu()
        if pre is not None and self.world > 1:
            self.__dict__["_pre_foreign"] = []
            gsc = g[:, n_main + per * psz + per:].reshape(self.world, per).cpu()
            for i in range(n_img):
                r, s = i % self.world, i // self.world
                if r == self.rank:
                    continue                                           # (filed by the hook)
                blk = g[r, n_main + s * psz:n_main + (s + 1) * psz]
                ent = (blk[: pcap * 2].view(1, pcap, 2), blk[pcap * 2:].view(1, pcap, 256), g[r, n_main + per * psz + s:n_main + per * psz + s + 1].view(torch.int32),
                       float(gsc[r, s]))
                pre._cache.pop((keys[i], quality), None)
                pre._cache[(keys[i], quality)] = ent
                self.__dict__.setdefault("_pre_foreign", []).append((pre, (keys[i], quality)))      # (views of the gathered buffer: dropped by release())
        dev_feats = []
        for i in range(n_img):
            r, s = i % self.world, i // self.world
            k = int(gc[r, s])
            size = np.array(np_shape(images[i])[:2], dtype=np.int32)
            t = gb[r, s, :k]
            dev_feats.append({"keypoints": t[:, 0:2], "descriptors_nd": t[:, 4:], "tile_idx": t[:, 3], "scores": t[:, 2], "image_size": size})
            if as_numpy:
                h = t.cpu().numpy()
                out.append({"keypoints": np.ascontiguousarray(h[:, 0:2]), "scores": np.ascontiguousarray(h[:, 2]), "tile_idx": np.ascontiguousarray(h[:, 3]),
                            "descriptors": np.ascontiguousarray(h[:, 4:].T), "image_size": size})
            else:
                out.append(dev_feats[-1])
        self._dev_feats, self._dev_token = dev_feats, out      # valid for exactly this list object; replaced by the next extract_all
        self.timings.update(extract_s=t1 - t0, feature_gather_s=time.perf_counter() - t1, feature_gather_bytes=int(flat.numel() * 4 * self.world))
        return out

    def release(self):
        """Drop the device tables of the last extract_all (they pin the all-gathered exchange buffer in HBM)."""
        self._dev_feats = self._dev_token = None
        for pre, key in self.__dict__.pop("_pre_foreign", []):
            pre._cache.pop(key, None)              # other ranks' preselection features are views of that buffer too (a miss re-derives them from the pixels)

    def _device_tables(self, feats: List[dict], dev) -> List[dict]:
        """Device feature tables for match_all: the exchange-buffer views when ``feats`` IS the list the last extract_all returned, else the
        caller's dicts uploaded (numpy keypoints (N,2), descriptors (D,N), tile_idx (N,)) — a filtered / modified / foreign list is never
        silently replaced by cached tables."""
        import numpy as np
        if self._dev_feats is not None and feats is self._dev_token:
            return self._dev_feats
        out = []
        for f in feats:
            if torch.is_tensor(f["keypoints"]):
                out.append(f)
                continue
            out.append({"keypoints": torch.from_numpy(np.ascontiguousarray(f["keypoints"], dtype=np.float32)).to(dev),
                        "descriptors_nd": torch.from_numpy(np.ascontiguousarray(np.asarray(f["descriptors"], dtype=np.float32).T)).to(dev),
                        "tile_idx": torch.from_numpy(np.ascontiguousarray(f["tile_idx"], dtype=np.float32)).to(dev),
                        "image_size": np.asarray(f["image_size"]).reshape(2)})
        return out

    # ---- phase 3a ---------------------------------------------------------------------------
    def _select(self, images, names, pairs, mine, n_tiles):
        """tile_selection of the image pairs ``mine`` -> {p: sorted [(t0, t1)]}.  PRESELECTION: the device preselector's LightGlue calls and
        vote kernels of all pairs are enqueued back to back under ONE range-guard read; the vote tables come back in one copy."""
        import numpy as np
        from . import capi
        from .tile_matching import get_size_by_quality, select_tile_pairs, tile_grid
        general = self.mat.config["general"]
        quality = getattr(general.get("quality", "HIGH"), "name", general.get("quality", "HIGH"))
        tile_size, overlap = general["tile_size"], general.get("tile_overlap", 0)
        min_matches = int(getattr(self.mat, "min_matches_per_tile", general.get("min_matches_per_tile", 5)))
        shape_only = lambda i: np.broadcast_to(np.float32(0), np_shape(images[i])[:2])
        grid = lambda i: tile_grid(get_size_by_quality(quality, np_shape(images[i])[:2]), tile_size, overlap)
        sel = {}
        if self.selection == "PRESELECTION" and len(mine):
            if general.get("preselection_pipeline", "superpoint+lightglue") != "superpoint+lightglue":
                raise ValueError("Only the superpoint+lightglue preselection pipeline is built on the MI355X path")
            pre = self.mat._preselector()
            dev = self._device()
            tmax = max(n_tiles)
            votes = torch.zeros(len(mine), tmax, tmax, dtype=torch.int32, device=dev)
            grids = {}

            def run():
                jobs, views = [], []
                for s, p in enumerate(mine):
                    a, b = int(pairs[p, 0]), int(pairs[p, 1])
                    for i in (a, b):
                        if i not in grids:
                            grids[i] = grid(i)
                    v = torch.empty(len(grids[a]), len(grids[b]), dtype=torch.int32, device=dev)
                    jobs.append((names[a], (lambda i=a: self._band(images, i)), names[b], (lambda i=b: self._band(images, i)), grids[a], grids[b], v))
                    views.append(v)
                pre.votes_device_many(jobs, tile_size, quality, pair_batch=self.preselection_pair_batch)   # one LightGlue stream over all the pairs
                for s, v in enumerate(views):
                    votes[s, : v.shape[0], : v.shape[1]] = v

            stream = ctypes_stream_of(dev)
            capi.run_guarded(pre.lib, stream, run, "tile preselection", "fallback")       # one guard read (and synchronisation) for the phase
            vh = votes.cpu().numpy().astype(np.int64)
            for s, p in enumerate(mine):
                a, b = int(pairs[p, 0]), int(pairs[p, 1])
                sel[p] = select_tile_pairs("PRESELECTION", list(grids[a]), list(grids[b]), vh[s, : len(grids[a]), : len(grids[b])], min_matches)
        else:
            needs_pixels = self.selection.startswith("PRESELECTION")
            for p in mine:
                a, b = int(pairs[p, 0]), int(pairs[p, 1])
                band = (lambda i: self._band(images, i)) if needs_pixels else shape_only
                sel[p] = self.mat.tile_selection(names[a], names[b], self.selection, image0=band(a), image1=band(b))
        if self.fallback is not None:
            for p in mine:
                if len(sel[p]) == 0:
                    self.n_fallback += 1
                    a, b = int(pairs[p, 0]), int(pairs[p, 1])
                    sel[p] = self.mat.tile_selection(names[a], names[b], self.fallback, image0=shape_only(a), image1=shape_only(b))
        return sel

    # ---- phases 3 + 4 ----------------------------------------------------------------------
    @torch.no_grad()
    def match_all(self, images: Sequence, feats: List[dict], pairs: torch.Tensor, names: Optional[Sequence[str]] = None) -> List:
        """pairs [P, 2] image indices.  Returns, on every rank, the list of (M, 2) int64 match arrays in the order of ``pairs``
        (the arrays MatcherBase._match_by_tile returns, MB:362-485)."""
        import time
        import numpy as np
        P = int(pairs.shape[0])
        dev = self._device()
        names = names if names is not None else [f"image{i:05d}" for i in range(len(feats))]
        n_tiles = [self._n_tiles(np_shape(im)) for im in images]
        tmax = max(n_tiles) if n_tiles else 1
        tables = self._device_tables(feats, dev)
        t0 = time.perf_counter()
        # ---- 3a: selection of the image pairs p = rank (mod world), then ONE small all-gather of the masks ----
        mine_sel = shard_indices(P, self.rank, self.world).tolist()
        per_sel = (P + self.world - 1) // self.world
        sel_mine = self._select(images, names, pairs, mine_sel, n_tiles)
        mask = torch.zeros(per_sel, tmax * tmax, dtype=torch.uint8)
        for s, p in enumerate(mine_sel):
            for (ta, tb) in sel_mine[p]:
                mask[s, ta * tmax + tb] = 1
        fb = torch.tensor([self.n_fallback], dtype=torch.int32)
        gm = _all_gather_cat(mask.to(dev)[None], self.world).cpu()                  # [world, per_sel, tmax^2]
        sel = []
        for p in range(P):
            idx = torch.nonzero(gm[p % self.world, p // self.world]).reshape(-1).tolist()
            sel.append([(k // tmax, k % tmax) for k in idx])                            # ascending = sorted (t0, t1): the reference's order
        t_sel = time.perf_counter()
        # ---- per-image tile counts (identical on every rank: every rank holds every table) and the cost-balanced deal ----
        counts = {}
        used = sorted({int(pairs[p, k]) for p in range(P) if sel[p] for k in (0, 1)})
        if used:
            from .tile_matching import device_tile_counts
            cdev = torch.stack([device_tile_counts(self.mat._lib, tables[i], tmax) for i in used])          # dim_op_tile_counts (csrc/sort_ops.hip)
            for i, c in zip(used, cdev.cpu().tolist()):                                  # ONE host read-back for the whole job
                counts[i] = c
        cost = torch.zeros(P, dtype=torch.float64)
        bound = torch.zeros(P, dtype=torch.int64)
        for p in range(P):
            a, b = int(pairs[p, 0]), int(pairs[p, 1])
            sel[p] = [(ta, tb) for ta, tb in sel[p] if counts[a][ta] > 0 and counts[b][tb] > 0]   # an empty tile cannot match (tile_matching.py)
            cost[p] = float(sum(counts[a][ta] * counts[b][tb] for ta, tb in sel[p]))
            bound[p] = sum(min(counts[a][ta], counts[b][tb]) for ta, tb in sel[p])
        rank_of, slot_of = balanced_shards(cost, self.world) if P else (torch.zeros(0, dtype=torch.long),) * 2
        mine = torch.nonzero(rank_of == self.rank).reshape(-1).tolist()
        per = (P + self.world - 1) // self.world
        cap_m = int(self.max_matches) if self.max_matches is not None else max(1, int(bound.max()) if P else 1)
        flat = torch.zeros(per + per * cap_m * 2, dtype=torch.int32, device=dev)
        cnt, rows = flat[:per], flat[per:].view(per, cap_m, 2)
        n_tp = sum(len(sel[p]) for p in mine)
        if n_tp:
            self._match_tile_pairs(tables, counts, pairs, sel, mine, cnt, rows, cap_m, tmax, dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        g = _all_gather_cat(flat[None], self.world)          # phase 4: ONE collective
        gc, gr = g[:, :per].cpu().numpy(), g[:, per:].reshape(self.world, per, cap_m, 2).cpu().numpy()
        out = []
        for p in range(P):
            r, s = int(rank_of[p]), int(slot_of[p])
            out.append(gr[r, s, : int(gc[r, s])].astype(np.int64))
        self.__dict__.pop("_band_cache", None)
        self.timings.update(match_s=t1 - t0, tile_selection_s=t_sel - t0, tile_matching_s=t1 - t_sel, match_gather_s=time.perf_counter() - t1,
                            match_gather_bytes=int(flat.numel() * 4 * self.world), selection_gather_bytes=int(mask.numel() * self.world),
                            tile_pairs_total=int(sum(len(x) for x in sel)), tile_pairs_this_rank=int(n_tp),
                            cost_this_rank=float(sum(float(cost[p]) for p in mine)), cost_total=float(cost.sum()))
        return out

    def _match_tile_pairs(self, tables, counts, pairs, sel, mine, cnt, rows, cap_m, tmax, dev):
        """The rank's tile pairs as ONE stream of dim_lg_match batches.  Per-image tile tables [tiles][cap_t][...] are gathered once per
        image; a tile pair is a row pair of the concatenated table; matches come back as tile-local indices and are mapped to merged-table
        indices through ``it``.  Per image pair: unique + lexicographic order (np.unique(axis=0), MB:462-474) = a sort of the 64-bit keys
        slot << 40 | idx0 << 20 | idx1."""
        from . import capi
        imgs = sorted({int(pairs[p, k]) for p in mine if sel[p] for k in (0, 1)})
        base = {i: j * tmax for j, i in enumerate(imgs)}
        cap_t = max(1, max(max(counts[i]) for i in imgs))
        T = len(imgs) * tmax
        D = int(tables[imgs[0]]["descriptors_nd"].shape[1])
        assert cap_t < (1 << 20) and max(int(tables[i]["keypoints"].shape[0]) for i in imgs) < (1 << 20) and len(mine) < (1 << 22)
        kt = torch.zeros(T, cap_t, 2, dtype=torch.float32, device=dev)
        dt = torch.zeros(T, cap_t, D, dtype=torch.float32, device=dev)
        it = torch.zeros(T, cap_t, dtype=torch.int64, device=dev)        # tile-local slot -> index in the image's merged table
        nt = torch.zeros(T, dtype=torch.int32, device=dev)
        st = torch.zeros(T, 2, dtype=torch.float32, device=dev)
        from .tile_matching import device_group_by_tile, device_unique_match_rows
        lib = self.mat._lib
        for i in imgs:        # keypoints grouped by tile, original order inside a tile (= the boolean-mask order of get_features_by_tile): dim_op_group_by_tile
            f = tables[i]
            rot = torch.arange(base[i], base[i] + tmax, dtype=torch.int32, device=dev)
            device_group_by_tile(lib, f, rot, cap_t, kt, dt, it, nt)
            st[base[i]: base[i] + tmax] = torch.as_tensor(np_f32(f["image_size"]), device=dev)
        tp = [(s, base[int(pairs[p, 0])] + ta, base[int(pairs[p, 1])] + tb) for s, p in enumerate(mine) for ta, tb in sel[p]]
        B = max(1, min(self.tile_pair_batch, len(tp)))
        # the handle is sized for the CONFIGURED batch, not for this job's: a job with fewer tile pairs than the batch (a warm-up pass, a small first image pair)
        # would otherwise leave a handle that the next longer job rebuilds — ~1 s of weight splitting, uploads and allocations inside its matching phase
        # (found in round 6: bench.py --workload config5 --tile-pair-batch 32 ran at 8.7 instead of 12.6 image-pairs/s with identical kernel time)
        net = self.mat._ensure_pairs(cap_t, max(1, self.tile_pair_batch))
        NK = net.nk
        pidx_all = torch.tensor([[r0, r1] for _, r0, r1 in tp], dtype=torch.int32, device=dev).contiguous()
        slot_all = torch.tensor([s for s, _, _ in tp], dtype=torch.int32, device=dev)
        keys = torch.empty(len(tp), NK, dtype=torch.int64, device=dev)
        stream = ctypes_stream_of(dev)

        def run():
            out = None
            for s0 in range(0, len(tp), B):
                pidx = pidx_all[s0:s0 + B].contiguous()
                b = int(pidx.shape[0])
                out = net.match_batch(kt, dt, nt, st, pair_idx=pidx, n_pairs=b, out=out)
                # tile-local match indices -> 64-bit keys  slot << 40 | idx0 << 20 | idx1  in the images' merged index space (~0: dead rows)
                capi.check(lib, lib.dim_op_tile_match_keys(capi.ptr(out["matches"]), capi.ptr(out["n_matches"]), capi.ptr(it), capi.ptr(pidx), capi.ptr(slot_all[s0:s0 + b]),
                                                           b, NK, cap_t, capi.ptr(keys[s0:s0 + b]), stream))

        _guarded(net, run, "tiled pipeline matching")        # ONE range-guard read for the phase; a re-run overwrites every key row
        # per image pair: unique rows in lexicographic order (np.unique(axis=0), MB:452-459) straight into the exchange buffer's row table
        n_full = torch.zeros(len(mine), dtype=torch.int32, device=dev)
        device_unique_match_rows(lib, keys, len(mine), cap_m, rows, cnt, n_full)
        over = int((n_full > cap_m).sum().item()) if self.max_matches is not None else 0     # only an explicit max_matches_per_pair can bind
        self.timings["truncated_pairs"] = over
        if over:
            logger.warning("TiledPairPipeline: %d image pair(s) have more unique matches than max_matches_per_pair=%d; the lists are cut to their first rows", over, cap_m)


def np_shape(image):
    import numpy as np
    return tuple(image.shape) if hasattr(image, "shape") else np.asarray(image).shape


def np_f32(x):
    import numpy as np
    return np.asarray(x, dtype=np.float32).reshape(2)


def ctypes_stream_of(dev):
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if torch.device(dev).type == "cuda" else None


def _band1(image):
    """the first band of an image array as float32 (what tile_selection reads with rasterio, MB:1021-1024)"""
    import numpy as np
    a = np.asarray(image)
    return np.ascontiguousarray(a if a.ndim == 2 else a[..., 0], dtype=np.float32)
