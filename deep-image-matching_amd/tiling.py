"""Tile-wise extraction for large-format images, batched on the GPU.

Host-side restatement of the reference's tiling glue —
``Tiler.compute_tiles_by_size`` (utils/tiling.py:62-192, with kornia 0.8.1's
``contrib.compute_padding`` restated since kornia is not a dependency here) and
``ExtractorBase._extract_by_tile`` (extractors/extractor_base.py:279-390) — with ONE difference:
the reference runs one network forward per tile, sequentially (EB:305-315); here all tiles of an
image (16 for the 6000x4000 / 1500x1000 case of BASELINE config 5) go through the resident
extractor as batches (``extract_batch``), and only the merge (origin shift, 2-px border mask,
``np.unique`` de-duplication) stays on the host, exactly as in the reference.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple, Union

import contextlib

import numpy as np
import torch


def compute_padding(original_size: Tuple[int, int], window_size: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """kornia.contrib.compute_padding (0.8.1) with stride = window, as DIM calls it
    (utils/tiling.py:124): (top, bottom, left, right) making (size - window) % window == 0."""
    pads = []
    for size, win in zip(original_size, window_size):
        rem = (size - win) % win
        pad = win - rem if rem != 0 else 0
        pads += [pad // 2, int(math.ceil(pad / 2))]
    return tuple(pads)


def compute_tiles_by_size(image: np.ndarray, window_size: Union[int, Tuple[int, int]], overlap: Union[int, Tuple[int, int]] = 0):
    """utils/tiling.py:62-192 (kornia != 0.7.1 branch).  image (H,W) or (H,W,C) numpy.  window_size /
    overlap are (x, y) when tuples.  Returns ({idx: tile (H_t,W_t,C)}, {idx: (x, y) origin in
    UNPADDED coordinates}, (top, bottom, left, right))."""
    if isinstance(window_size, int):
        win = (window_size, window_size)
    elif isinstance(window_size, (tuple, list)):
        win = (window_size[1], window_size[0])  # -> (H, W)
    else:
        raise TypeError("window_size must be an integer or a tuple of integers")
    if isinstance(overlap, int):
        ov = (overlap, overlap)
    elif isinstance(overlap, (tuple, list)):
        ov = (overlap[1], overlap[0])
    else:
        raise TypeError("overlap must be an integer or a tuple of integers")
    if not isinstance(image, np.ndarray):
        raise TypeError("input must be a numpy array")
    img = image if image.ndim == 3 else image[..., None]
    H, W = img.shape[:2]
    pad = compute_padding((H, W), win)
    stride = (win[0] - ov[0], win[1] - ov[1])
    padded = np.pad(img, ((pad[0], pad[1]), (pad[2], pad[3]), (0, 0)), mode="constant", constant_values=0)
    ph, pw = padded.shape[:2]
    tiles, k = {}, 0
    for y in range(0, ph - win[0] + 1, stride[0]):
        for x in range(0, pw - win[1] + 1, stride[1]):
            tiles[k] = padded[y:y + win[0], x:x + win[1]]
            k += 1
    n_rows = (H + pad[0] + pad[1] - win[0]) // stride[0] + 1
    n_cols = (W + pad[2] + pad[3] - win[1]) // stride[1] + 1
    origins = {r * n_cols + c: (-pad[2] + c * stride[1], -pad[0] + r * stride[0]) for r in range(n_rows) for c in range(n_cols)}
    return tiles, origins, pad


def merge_tile_features(per_tile: Dict[int, dict], origins: Dict[int, Tuple[int, int]], image_shape, descriptor_size: int,
                        select_unique: bool = True) -> dict:
    """EB:330-390: shift to image coordinates, drop keypoints within 2 px of the (unpadded) image
    border, concatenate, de-duplicate with np.unique (which re-sorts lexicographically by (x, y))."""
    kl, dl, sl, tl = [], [], [], []
    for idx, f in per_tile.items():
        kp = f["keypoints"] + np.array(origins[idx], dtype=f["keypoints"].dtype)
        thr = 2
        m = (kp[:, 0] >= thr) & (kp[:, 0] < image_shape[1] - thr) & (kp[:, 1] >= thr) & (kp[:, 1] < image_shape[0] - thr)
        kp = kp[m]
        if len(kp) > 0:
            kl.append(kp); dl.append(f["descriptors"][:, m]); sl.append(f["scores"][m]); tl.append(np.full(len(kp), idx, dtype=np.float32))
    if kl:
        kpts, desc, scores, tidx = np.vstack(kl), np.hstack(dl), np.concatenate(sl), np.concatenate(tl)
    else:
        kpts = np.array([], dtype=np.float32).reshape(0, 2)
        desc = np.array([], dtype=np.float32).reshape(descriptor_size, 0)
        scores, tidx = np.array([], dtype=np.float32), np.array([], dtype=np.float32)
    if select_unique:
        kpts, u = np.unique(kpts, axis=0, return_index=True)
        desc, tidx, scores = desc[:, u], tidx[u], scores[u]
    return {"keypoints": kpts, "descriptors": desc, "scores": scores, "tile_idx": tidx}


DEVICE_MERGE_MAX_SLOTS = 1 << 18   # 262 144 slots = 6.9e10 rank compares (~10 ms); above it the host merge is faster


def merge_tile_features_device(lib, dev, stream, tables, origins_xy, tile_ids, image_shape, select_unique: bool = True, as_device: bool = False) -> dict:
    """merge_tile_features on the device (csrc/tile_merge.hip: shift, border filter, np.unique's lexicographic order and
    first-occurrence rule, descriptor transpose) from the extractor's per-chunk device tables (kp [T,cap,2], scores [T,cap],
    desc [T,cap,D], n [T]); ONE device-to-host copy of the final arrays.  EB:330-390."""
    import contextlib
    import ctypes
    from . import capi
    if len(tables) == 1:
        kp, sc, de, n = tables[0]
    else:  # more tiles than one extractor batch: one table (capacities can differ after a keep-all regrow)
        cap = max(int(t[0].shape[1]) for t in tables)
        def pad(x):
            if x.shape[1] == cap:
                return x
            y = x.new_zeros((x.shape[0], cap) + tuple(x.shape[2:]))
            y[:, :x.shape[1]] = x
            return y
        kp, sc, de = (torch.cat([pad(t[k]) for t in tables]) for k in range(3))
        n = torch.cat([t[3] for t in tables])
    kp, sc, de, n = kp.contiguous(), sc.contiguous(), de.contiguous(), n.to(torch.int32).contiguous()
    T, cap, D = int(de.shape[0]), int(de.shape[1]), int(de.shape[2])
    assert T == len(tile_ids) == len(origins_xy)
    if T * cap > DEVICE_MERGE_MAX_SLOTS:
        # tm_rank_kernel ranks all T * cap slots against each other (O(n^2) compares, dead padding slots included): fine at 16 tiles x
        # 4096 slots (~1 ms), quadratic beyond — keep-all capacities or images with hundreds of tiles go through the host merge
        # (the reference's own numpy statements), which only touches the live keypoints
        n_h = n.cpu().numpy()
        kp_h, sc_h, de_h = kp.cpu().numpy(), sc.cpu().numpy(), de.cpu().numpy()
        per_tile = {tid: {"keypoints": kp_h[i, :n_h[i]], "scores": sc_h[i, :n_h[i]], "descriptors": de_h[i, :n_h[i]].T} for i, tid in enumerate(tile_ids)}
        res = merge_tile_features(per_tile, {tid: tuple(origins_xy[i]) for i, tid in enumerate(tile_ids)}, image_shape, D, select_unique)
        return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in res.items()} if as_device else res
    og = torch.tensor(origins_xy, dtype=torch.int32, device=dev).reshape(T, 2).contiguous()
    ids = torch.tensor([float(i) for i in tile_ids], dtype=torch.float32, device=dev)
    lib.dim_op_merge_tiles_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(int(lib.dim_op_merge_tiles_workspace_bytes(T, cap)), dtype=torch.uint8, device=dev)
    rows = T * cap
    o_kp = torch.empty(rows, 2, dtype=torch.float32, device=dev)
    o_sc = torch.empty(rows, dtype=torch.float32, device=dev)
    o_ti = torch.empty(rows, dtype=torch.float32, device=dev)
    o_de = torch.empty(rows * D, dtype=torch.float32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
        capi.check(lib, lib.dim_op_merge_tiles(capi.ptr(kp), capi.ptr(sc), capi.ptr(de), capi.ptr(n), capi.ptr(og), capi.ptr(ids), T, cap, D,
                                               int(image_shape[0]), int(image_shape[1]), int(bool(select_unique)), capi.ptr(ws),
                                               capi.ptr(o_kp), capi.ptr(o_sc), capi.ptr(o_ti), capi.ptr(o_de), capi.ptr(n_out), stream))
    N = int(n_out.item())
    if as_device:   # the merged table stays in HBM (pipeline.TiledPairPipeline packs it into its exchange buffer there)
        return {"keypoints": o_kp[:N], "descriptors": o_de[:D * N].reshape(D, N), "scores": o_sc[:N], "tile_idx": o_ti[:N]}
    return {"keypoints": o_kp[:N].cpu().numpy(), "descriptors": o_de[:D * N].reshape(D, N).cpu().numpy(),
            "scores": o_sc[:N].cpu().numpy(), "tile_idx": o_ti[:N].cpu().numpy()}


class BatchedTilingMixin:
    """Overrides ExtractorBase._extract_by_tile with the batched version.  Needs ``self._net``
    (SuperPointHIP / AlikedHIP built by ``self._ensure_batch``), ``self.grayscale`` and
    ``self.descriptor_size``."""

    tile_batch = 16

    @torch.no_grad()
    def _extract_by_tile(self, image: np.ndarray, select_unique: bool = True, as_device: bool = False, on_device_image=None) -> dict:
        """The image goes to the device ONCE (one H2D copy of the caller's array); zero padding, tile slicing and
        _frame2tensor's /255 happen there (IEEE fp32 division: bit-identical to the host's), so no host pass touches
        the 288 MB of a 6000x4000 RGB float image.  ``on_device_image(src)``: called with the device copy ([H, W] or [H, W, C] float32, 0..255) while
        it exists — the tiled pipeline derives the tile-preselection features of the image from it instead of touching the host array again."""
        general = self.config["general"]
        win, ov = general["tile_size"], general.get("tile_overlap", 0)
        win_hw = (win, win) if isinstance(win, int) else (win[1], win[0])
        ov_hw = (ov, ov) if isinstance(ov, int) else (ov[1], ov[0])
        if not isinstance(image, np.ndarray):
            raise TypeError("input must be a numpy array")
        H, W = image.shape[:2]
        pad = compute_padding((H, W), win_hw)
        stride = (win_hw[0] - ov_hw[0], win_hw[1] - ov_hw[1])
        n_rows = (H + pad[0] + pad[1] - win_hw[0]) // stride[0] + 1
        n_cols = (W + pad[2] + pad[3] - win_hw[1]) // stride[1] + 1
        origins = {r * n_cols + c: (-pad[2] + c * stride[1], -pad[0] + r * stride[0]) for r in range(n_rows) for c in range(n_cols)}
        th, tw = win_hw
        net = self._ensure_batch(th, tw, self.tile_batch)
        dev = net.device
        src = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(dev)
        C = 1 if src.dim() == 2 else int(src.shape[2])
        if on_device_image is not None:
            on_device_image(src)
        idxs = sorted(origins)
        tables = []
        from . import capi
        import ctypes
        lib = net.lib
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        for s in range(0, len(idxs), self.tile_batch):
            chunk = idxs[s:s + self.tile_batch]
            # dim_op_gather_tiles_f32: zero padding, tile slicing and _frame2tensor's / 255 in one pass from the image as it arrived
            og = torch.tensor([origins[i] for i in chunk], dtype=torch.int32, device=dev).contiguous()
            t = torch.empty((len(chunk), th, tw) if self.grayscale and C == 1 else (len(chunk), th, tw, C), dtype=torch.float32, device=dev)
            with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
                capi.check(lib, lib.dim_op_gather_tiles_f32(capi.ptr(src), H, W, C, capi.ptr(og), len(chunk), th, tw, capi.ptr(t), 1, stream))
            if self.grayscale and C != 1:
                t = t[..., 0].contiguous()   # a colour array handed to a grey extractor: channel 0, as stack[..., 0] did
            # under the fp16x3 range guard; SuperPoint in keep-all mode also repeats a batch that overflowed its slots
            run = getattr(net, "extract_batch_guarded", net.extract_batch)
            kp, sc, de, n = run(t)
            if hasattr(self, "_regrow") and self._regrow(net, len(chunk)):
                net = self._ensure_batch(th, tw, self.tile_batch)
                kp, sc, de, n = getattr(net, "extract_batch_guarded", net.extract_batch)(t)
            tables.append((kp, sc, de, n))
        return merge_tile_features_device(lib, dev, stream, tables, [origins[i] for i in idxs], idxs, image.shape, select_unique, as_device)
