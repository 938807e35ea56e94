"""Data-parallel extraction + matching over the GPUs of one node.

The reference is single-process / single-device (image_matching.py:413-494: one
``extract`` per image, one ``match`` per pair).  Here one process drives one GPU and

  phase 1  images  i ≡ rank (mod world) are extracted locally (batched dim_sp_extract),
  phase 2  ONE all-gather of the fixed-slot feature tables makes every rank hold all
           features (n_img x cap x (2 + 1 + D) floats + n_img counts),
  phase 3  the pair list (``itertools.combinations`` order for bruteforce,
           pairs_generator.py:37-38) is sharded round-robin, each rank matches its shard
           in batches (dim_lg_match with a pair-index table: no feature copies),
  phase 4  ONE all-gather of the per-rank match tables (counts + padded (idx0, idx1)
           rows + scores) gives every rank the complete result.

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

    # ---- phases 1+2 ------------------------------------------------------------------------
    @torch.no_grad()
    def extract_all(self, images: torch.Tensor, image_sizes: Optional[torch.Tensor] = None):
        """images [n_img, H, W] float32 in [0,1], identical on every rank (or at least the
        rank's own shard valid).  Returns the GLOBAL feature table (kpts [n_img,cap,2],
        scores [n_img,cap], desc [n_img,cap,D], n [n_img], size [n_img,2]) on every rank."""
        n_img, H, W = images.shape
        mine = shard_indices(n_img, self.rank, self.world)
        per = (n_img + self.world - 1) // self.world
        cap, dev = self.ext.capacity, images.device
        kp = torch.zeros(per, cap, 2, dtype=torch.float32, device=dev)
        sc = torch.zeros(per, cap, dtype=torch.float32, device=dev)
        de = torch.zeros(per, cap, 256, dtype=torch.float32, device=dev)
        n = torch.zeros(per, dtype=torch.int32, device=dev)
        B = self.ext.max_batch

        def run():  # every batch of the shard is enqueued back to back; the fp16x3 range guard is read once per phase
            for s in range(0, len(mine), B):
                idx = mine[s:s + B]
                k_, s_, d_, n_ = self.ext.extract_batch(images[idx.to(dev)].contiguous())
                kp[s:s + len(idx)], sc[s:s + len(idx)], de[s:s + len(idx)], n[s:s + len(idx)] = k_, s_, d_, n_

        _guarded(self.ext, run, "pipeline extraction")
        # phase 2: one all-gather per table; slot (r, j) holds image j*world + r
        kp_g, sc_g, de_g, n_g = (_all_gather_cat(t, self.world) for t in (kp, sc, de, n))
        if self.world > 1:
            order = torch.arange(self.world * per, device=dev).reshape(self.world, per).t().reshape(-1)[:n_img]
            kp_g, sc_g, de_g, n_g = kp_g[order], sc_g[order], de_g[order], n_g[order]
        else:
            kp_g, sc_g, de_g, n_g = kp_g[:n_img], sc_g[:n_img], de_g[:n_img], n_g[:n_img]
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
        kp, sc, de, n, size = table
        dev = kp.device
        P = pairs.shape[0]
        mine = shard_indices(P, self.rank, self.world)
        per = (P + self.world - 1) // self.world
        NK, B = self.mat.nk, self.mat.max_pairs
        cnt = torch.zeros(per, dtype=torch.int32, device=dev)
        mt = torch.zeros(per, NK, 2, dtype=torch.int64, device=dev)
        ms = torch.zeros(per, NK, dtype=torch.float32, device=dev)
        stp = torch.zeros(per, dtype=torch.int32, device=dev)
        prn = torch.zeros(per, 2, NK, dtype=torch.int32, device=dev) if aux else None
        my_pairs = pairs[mine].to(dev, torch.int32).contiguous()

        def run():
            for s in range(0, len(mine), B):
                pp = my_pairs[s:s + B].contiguous()
                o = self.mat.match_batch(kp, de, n, size, pair_idx=pp)
                b = pp.shape[0]
                live = torch.arange(NK, device=dev)[None, :] < o["n_matches"][:, None]  # rows beyond n_matches are unspecified
                cnt[s:s + b] = o["n_matches"]
                mt[s:s + b] = torch.where(live[..., None], o["matches"], torch.zeros_like(o["matches"]))
                ms[s:s + b] = torch.where(live, o["scores"], torch.zeros_like(o["scores"]))
                stp[s:s + b] = o["stop"]
                if aux:
                    prn[s:s + b] = o["prune01"]

        _guarded(self.mat, run, "pipeline matching")
        cnt_g, mt_g, ms_g = (_all_gather_cat(t, self.world) for t in (cnt, mt, ms))
        if self.world > 1:
            order = torch.arange(self.world * per, device=dev).reshape(self.world, per).t().reshape(-1)[:P]
            cnt_g, mt_g, ms_g = cnt_g[order], mt_g[order], ms_g[order]
        else:
            cnt_g, mt_g, ms_g = cnt_g[:P], mt_g[:P], ms_g[:P]
        if aux:
            stp_g, prn_g = _all_gather_cat(stp, self.world), _all_gather_cat(prn, self.world)
            if self.world > 1:
                stp_g, prn_g = stp_g[order], prn_g[order]
            return cnt_g, mt_g, ms_g, stp_g[:P], prn_g[:P]
        return cnt_g, mt_g, ms_g

    @staticmethod
    def to_match_lists(cnt: torch.Tensor, mt: torch.Tensor, ms: torch.Tensor) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """Host-side unpadding: [(matches (S,2) int64, scores (S,)) per pair] — the arrays DIM
        writes to raw_matches.h5 (matcher_base.py:282-285)."""
        cnt = cnt.cpu()
        mt, ms = mt.cpu(), ms.cpu()
        return [(mt[p, : int(cnt[p])], ms[p, : int(cnt[p])]) for p in range(cnt.shape[0])]
