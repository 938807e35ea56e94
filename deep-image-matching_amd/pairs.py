"""Pair generation on the device: ``matching_lowres`` (the reference's default CLI strategy).

Restatement of ``pairs_generator.py``: ``pairs_from_sequential`` (:22-34), ``pairs_from_bruteforce``
(:37-38) and the SuperPoint branch of ``pairs_from_lowres`` (:41-235, the only branch that runs —
``use_superpoint = True`` is hard-wired at :95): every image is read as grey, down-sampled with
``cv2.resize(..., INTER_AREA)`` to ``resize_max`` on its long side, described by hloc's SuperPoint wrapper
(nms 3 / 2048 keypoints / threshold 0.0005 + the wrapper's ``fix_sampling=True``, Q3) and every image pair
is matched with LightGlue (7 layers, depth 0.9 / width 0.95 / filter 0.3, no ``image_size``: keypoint
extent); a pair is kept when it has more than ``min_matches`` matches (:222-223).

The reference runs N extractions and N(N-1)/2 single-pair matcher calls, moving every feature set to
the host and back.  Here the down-sampling, the extractions and all matches stay in HBM: the features
form one device table and the pairs go through ``LightGlueHIP.match_batch`` in batches (pair -> row
indirection); only the per-pair match counts come back.  With ``rank`` / ``world`` the pairs are sharded
like ``pipeline.PairMatchingPipeline`` does and the (disjoint) per-rank counts are summed with one all-reduce.

``do_geometric_verification`` (cv2 RANSAC on the low-res matches, :206-220) is not rebuilt.
"""
from __future__ import annotations

import ctypes
from itertools import combinations
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi
from .lightglue_hip import LightGlueHIP
from .superpoint_hip import SuperPointHIP

# pairs_generator.py:103-126
LOWRES_SP_CONF = {"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.0005, "remove_borders": 4, "fix_sampling": True}
LOWRES_LG_CONF = {"n_layers": 7, "depth_confidence": 0.9, "width_confidence": 0.95, "filter_threshold": 0.3}


def pairs_from_sequential(img_list: Sequence, overlap: int) -> List[tuple]:
    pairs = []
    for i in range(len(img_list)):
        for k in range(overlap):
            j = i + k + 1
            if j >= len(img_list):
                break
            pairs.append((img_list[i], img_list[j]))
    return pairs


def pairs_from_bruteforce(img_list: Sequence) -> List[tuple]:
    return list(combinations(img_list, 2))


def pairs_from_retrieval(query_names: Sequence[str], db_names: Sequence[str], query_desc, db_desc, num_matched: int,
                         min_score: Optional[float] = 0.0, device="cuda", lib=None) -> List[tuple]:
    """thirdparty/hloc/pairs_from_retrieval.py:49-70,108-112 on the device: global descriptors (N, D) -> similarity GEMM
    -> self / low-score masking -> top-``num_matched`` per query; returns (query, db) name pairs in the reference's order
    (query by query, best first).  Only the (nq, k) index table leaves the device."""
    lib = lib if lib is not None else capi.load()
    dev = torch.device(device)
    q = torch.as_tensor(np.asarray(query_desc), dtype=torch.float32)
    d = torch.as_tensor(np.asarray(db_desc), dtype=torch.float32)
    nq, nd, D = q.shape[0], d.shape[0], q.shape[1]
    if nq == 0 or nd == 0:
        return []
    pad = (-D) % 32
    if pad:
        q, d = torch.nn.functional.pad(q, (0, pad)), torch.nn.functional.pad(d, (0, pad))
    k = min(int(num_matched), nd)   # torch.topk raises for k > nd; hloc callers keep num_matched <= number of images
    q, d = q.contiguous().to(dev), d.contiguous().to(dev)
    invalid = torch.from_numpy(np.array(query_names)[:, None] == np.array(db_names)[None]).to(torch.uint8).contiguous().to(dev)
    sim = torch.empty(nq, nd, dtype=torch.float32, device=dev)
    idx = torch.empty(nq, k, dtype=torch.int32, device=dev)
    val = torch.empty(nq, k, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
    import contextlib
    with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):   # launch on dev, not on the current device
        capi.check(lib, lib.dim_op_retrieval_topk(capi.ptr(q), nq, capi.ptr(d), nd, D + pad, capi.ptr(invalid), k,
                                                  ctypes.c_float(0.0 if min_score is None else float(min_score)), int(min_score is not None),
                                                  capi.ptr(sim), capi.ptr(idx), capi.ptr(val), stream))
    idx = idx.cpu().numpy()
    return [(query_names[i], db_names[j]) for i in range(nq) for j in idx[i] if j >= 0]


class LowresPairSelector:
    """``pairs_from_lowres`` with resident networks.  ``images``: grey float32 arrays (0..255) in list order."""

    def __init__(self, sp_state_dict, lg_state_dict, resize_max: int = 1000, min_matches: int = 20, pair_batch: int = 8,
                 device="cuda", lib=None, rank: int = 0, world: int = 1):
        self.resize_max, self.min_matches, self.pair_batch = int(resize_max), int(min_matches), int(pair_batch)
        self.device = torch.device(device)
        self.lib = lib if lib is not None else capi.load()
        self.rank, self.world = rank, world
        self._sp_sd, self._lg_sd = sp_state_dict, lg_state_dict
        self._sp: Optional[SuperPointHIP] = None
        self._sp_hw = (0, 0)
        self._lg: Optional[LightGlueHIP] = None

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def downsample(self, image: np.ndarray) -> torch.Tensor:
        """pairs_generator.py:141-146: size (w, h), scale = resize_max / max, rounded size, INTER_AREA, /255."""
        H, W = image.shape[:2]
        scale = self.resize_max / max(W, H)
        w, h = int(round(W * scale)), int(round(H * scale))
        src = torch.as_tensor(np.ascontiguousarray(image, dtype=np.float32)).to(self.device)
        dst = torch.empty(h, w, dtype=torch.float32, device=self.device)
        capi.check(self.lib, self.lib.dim_op_resize_area_f32(capi.ptr(src), H, W, capi.ptr(dst), h, w, 1, self._stream()))
        return dst

    def extract(self, images: Sequence[np.ndarray]):
        """Feature table of the down-sampled images: kpts [n, cap, 2], desc [n, cap, 256], n_kpts [n], size [n, 2]."""
        cap = 2048
        n = len(images)
        kt = torch.zeros(n, cap, 2, dtype=torch.float32, device=self.device)
        dt = torch.zeros(n, cap, 256, dtype=torch.float32, device=self.device)
        nt = torch.zeros(n, dtype=torch.int32, device=self.device)
        st = torch.ones(n, 2, dtype=torch.float32, device=self.device)
        for i, im in enumerate(images):
            small = self.downsample(im)
            h, w = small.shape
            if self._sp is None or h > self._sp_hw[0] or w > self._sp_hw[1]:
                self._sp_hw = (max(h, self._sp_hw[0], self.resize_max), max(w, self._sp_hw[1], self.resize_max))
                self._sp = SuperPointHIP(self._sp_sd, LOWRES_SP_CONF, max_batch=1, max_hw=self._sp_hw, capacity=cap,
                                         device=self.device, lib=self.lib)
            kp, _, de, nk = self._sp.extract_batch_guarded(small[None].contiguous())
            kt[i], dt[i], nt[i] = kp[0], de[0], nk[0]
            k = kp[0, : int(nk[0].item())]
            if k.numel():  # no image_size in the reference's call: LightGlue uses the keypoint extent (LGN:26-27)
                st[i] = 1 + k.max(0).values - k.min(0).values
        return kt, dt, nt, st

    def match_counts(self, table, pairs: Sequence[Tuple[int, int]]) -> np.ndarray:
        """len(matches) of every (i, j) in ``pairs`` (this rank's shard when world > 1, then all-gathered)."""
        kt, dt, nt, st = table
        if self._lg is None:
            self._lg = LightGlueHIP(self._lg_sd, LOWRES_LG_CONF, max_pairs=self.pair_batch, max_kpts=kt.shape[1], device=self.device, lib=self.lib)
        P = len(pairs)
        mine = list(range(self.rank, P, self.world))
        counts = torch.zeros(P, dtype=torch.int32, device=self.device)

        def run():  # all chunks are enqueued back to back; the fp16x3 range guard is read once for the whole shard
            out = None
            for s in range(0, len(mine), self.pair_batch):
                chunk = mine[s:s + self.pair_batch]
                pidx = torch.tensor([pairs[c] for c in chunk], dtype=torch.int32, device=self.device).contiguous()
                out = self._lg.match_batch(kt, dt, nt, st, pair_idx=pidx, n_pairs=len(chunk), out=out)  # the first chunk is the largest
                counts[torch.tensor(chunk, device=self.device)] = out["n_matches"][: len(chunk)]

        with self._lg._ctx():
            capi.run_guarded(self.lib, self._stream(), run, "matching_lowres", self._lg.on_saturation, handle=self._lg._h, arithmetic=self._lg.arithmetic)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)  # shards are disjoint: the sum is the gather
        return counts.cpu().numpy()

    def select(self, names: Sequence, images: Sequence[np.ndarray]) -> List[tuple]:
        """pairs_from_lowres: the brute-force pairs (combinations order) with more than min_matches matches."""
        idx_pairs = list(combinations(range(len(names)), 2))
        if not idx_pairs:
            return []
        counts = self.match_counts(self.extract(images), idx_pairs)
        return [(names[i], names[j]) for (i, j), c in zip(idx_pairs, counts) if c > self.min_matches]
