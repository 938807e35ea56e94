"""Seeded inputs / weights of the golden cases.  Shared by oracle/make_golden.py (which
records the REFERENCE's outputs into tests/golden/*.npz) and by the tests (which
regenerate the inputs from the same seeds)."""
from __future__ import annotations

import importlib
import math

import torch

weights = importlib.import_module("deep-image-matching_amd.weights")

SP_CASES = {
    # zoo config (config.py:94-99) scaled down: top-k is exercised (noise gives >> k maxima)
    "noise_topk": {"seed": 0, "H": 96, "W": 128, "kind": "noise", "wseed": 1234,
                   "cfg": {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 64, "remove_borders": 4}},
    # shipped YAML config (nms 4 / thr 0.005), unlimited keypoints, sides not multiples of 8 (Q12)
    "blobs_all": {"seed": 1, "H": 77, "W": 102, "kind": "blobs", "wseed": 1234,
                  "cfg": {"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": -1, "remove_borders": 4}},
    # DIM/hloc fixed sampler (Q3)
    "noise_fixsampling": {"seed": 2, "H": 64, "W": 80, "kind": "noise", "wseed": 99,
                          "cfg": {"nms_radius": 2, "keypoint_threshold": 0.001, "max_keypoints": 100, "remove_borders": 2,
                                  "fix_sampling": True}},
}


def sp_weights(case):
    return weights.synthetic_superpoint_state_dict(case["wseed"])


def sp_image(case) -> torch.Tensor:
    g = torch.Generator().manual_seed(case["seed"])
    H, W = case["H"], case["W"]
    if case["kind"] == "noise":
        return torch.rand(1, 1, H, W, generator=g)
    # sum of random Gaussian blobs + a little noise: realistic sparsity for NMS/threshold
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    img = torch.zeros(H, W)
    for _ in range(40):
        cy, cx = torch.rand(2, generator=g) * torch.tensor([H, W], dtype=torch.float32)
        s = 1.0 + 4.0 * torch.rand(1, generator=g)
        a = torch.rand(1, generator=g)
        img += a * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    img = img / img.max() * 0.9 + 0.05 * torch.rand(H, W, generator=g)
    return img.clamp(0, 1)[None, None].contiguous()


LG_CASES = {
    # reference defaults (depth .95 / width .99), non-square image_size as DIM feeds it (H, W) (Q4)
    "default": {"seed": 10, "m": 96, "n": 80, "input_dim": 256, "wseed": 0, "gain": 2.0, "heads_gain": 1.0,
                "size0": (480.0, 640.0), "size1": (618.0, 640.0),
                "conf": {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}},
    # fixed work: no early stop, no pruning (the benchmark mode)
    "fixed": {"seed": 11, "m": 70, "n": 130, "input_dim": 256, "wseed": 1, "gain": 2.0, "heads_gain": 1.0,
              "size0": (1024.0, 1024.0), "size1": (1024.0, 1024.0),
              "conf": {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}},
    # saturated confidence/matchability heads: pruning removes points every layer, early stop fires
    "adaptive": {"seed": 12, "m": 150, "n": 110, "input_dim": 256, "wseed": 2, "gain": 2.0, "heads_gain": 30.0,
                 "size0": (768.0, 1024.0), "size1": (1024.0, 768.0),
                 "conf": {"depth_confidence": 0.5, "width_confidence": 0.99, "filter_threshold": 0.0}},
    # pruning removes points over several layers (prune counters 3..9), all 9 layers run, 19 matches survive
    # (r1's wseed 6 / match_gain 2.5 pruned so hard that the match list was empty: vacuous for matches / scores)
    "prune_only": {"seed": 12, "m": 150, "n": 110, "input_dim": 256, "wseed": 14, "gain": 2.0, "match_gain": 1.0,
                   "size0": (768.0, 1024.0), "size1": (1024.0, 768.0),
                   "conf": {"depth_confidence": -1, "width_confidence": 0.99, "filter_threshold": 0.0}},
    # everything gets pruned away: the reference's "no keypoints" exit (LGN:518-540)
    "prune_to_empty": {"seed": 12, "m": 150, "n": 110, "input_dim": 256, "wseed": 4, "gain": 2.0, "match_gain": 30.0,
                       "size0": (768.0, 1024.0), "size1": (1024.0, 768.0),
                       "conf": {"depth_confidence": -1, "width_confidence": 0.99, "filter_threshold": 0.0}},
    # ALIKED-style 128-d descriptors through input_proj, the reference's default threshold 0.1: early stop at layer 4,
    # 12 matches above the threshold (r1's wseed 3 / heads_gain 5 gave none)
    "aliked_dim": {"seed": 13, "m": 64, "n": 64, "input_dim": 128, "wseed": 6, "gain": 2.0, "heads_gain": 2.0,
                   "size0": (512.0, 768.0), "size1": (512.0, 768.0),
                   "conf": {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}},
    # 128-d inputs with pruning AND early stop AND surviving matches (prune counters 2..6, stop 6, 4 matches)
    "aliked_prune": {"seed": 13, "m": 64, "n": 64, "input_dim": 128, "wseed": 13, "gain": 2.0, "heads_gain": 2.0,
                     "size0": (512.0, 768.0), "size1": (512.0, 768.0),
                     "conf": {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}},
}


def lg_weights(case):
    sd = weights.synthetic_lightglue_state_dict(case["wseed"], case["input_dim"], gain=case["gain"])
    hg = case.get("heads_gain", 1.0)
    mg = case.get("match_gain", hg)
    tg = case.get("token_gain", hg)
    for k in sd:
        if k.endswith("weight"):
            if "matchability" in k:
                sd[k] = sd[k] * mg
            if "token_confidence" in k:
                sd[k] = sd[k] * tg
    return sd


def lg_inputs(case):
    g = torch.Generator().manual_seed(case["seed"])
    out = []
    # image 1 = a perturbed, shuffled copy of part of image 0 so that real correspondences exist
    m, n, D = case["m"], case["n"], case["input_dim"]
    (h0, w0), (h1, w1) = case["size0"], case["size1"]
    k0 = torch.rand(m, 2, generator=g) * torch.tensor([w0, h0])
    d0 = torch.nn.functional.normalize(torch.randn(m, D, generator=g), dim=-1)
    perm = torch.randperm(max(m, n), generator=g)[:n] % m
    k1 = (k0[perm] / torch.tensor([w0, h0]) * torch.tensor([w1, h1]) + torch.randn(n, 2, generator=g) * 2.0)
    d1 = torch.nn.functional.normalize(d0[perm] + 0.3 * torch.randn(n, D, generator=g) / math.sqrt(D) * 4, dim=-1)
    out.append({"kpts": k0.contiguous(), "desc": d0.contiguous(), "size": torch.tensor([h0, w0])})
    out.append({"kpts": k1.contiguous(), "desc": d1.contiguous(), "size": torch.tensor([h1, w1])})
    return out


AL_CASES = {
    # RGB, sides not multiples of 32 (InputPadder replicate padding), DIM's default config scaled down
    "rgb_pad": {"seed": 21, "H": 70, "W": 100, "C": 3, "wseed": 7,
                "cfg": {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 2}},
    # gray input repeated to RGB, n_limit binding (more maxima than max_num_keypoints), radius 3
    "gray_limit": {"seed": 22, "H": 64, "W": 96, "C": 1, "wseed": 8,
                   "cfg": {"model_name": "aliked-n16rot", "max_num_keypoints": 60, "detection_threshold": 0.2, "nms_radius": 3}},
    # aliked-n32 (ALN:577): 32 instead of 16 deformable sample positions in the descriptor head
    "n32": {"seed": 23, "H": 72, "W": 88, "C": 3, "wseed": 9,
            "cfg": {"model_name": "aliked-n32", "max_num_keypoints": 300, "detection_threshold": 0.2, "nms_radius": 2}},
    # aliked-t16 (ALN:574): 8 / 16 / 32 / 64 channels, 64-d descriptors; gray input, padding on both sides
    "t16": {"seed": 24, "H": 80, "W": 76, "C": 1, "wseed": 10,
            "cfg": {"model_name": "aliked-t16", "max_num_keypoints": 300, "detection_threshold": 0.2, "nms_radius": 2}},
}


def al_weights(case):
    return weights.synthetic_aliked_state_dict(case["wseed"], case["cfg"]["model_name"])


def al_image(case) -> torch.Tensor:
    g = torch.Generator().manual_seed(case["seed"])
    return torch.rand(1, case["C"], case["H"], case["W"], generator=g)


# ---- BASELINE configs[0] on its REAL inputs ---------------------------------------------------------------
# tests/assets/config1/ holds byte copies of the photographs the reference ships: assets/example_sacre_coeur/images/*.jpg (the five images of
# configs[0]) and assets/pytest/images/DSC_646{6,7,8}.jpg (the reference's own pytest fixture, tests/conftest.py:11-37).  Decoding: the
# reference reads with rasterio (GDAL's libjpeg); neither exists here nor on the GPU box, so PIL decodes — the goldens record a digest of the
# decoded pixels and the tests refuse to compare against goldens made from different pixels (decoder differences are outside the parity claim,
# SURVEY 8(c)/(d)).
from pathlib import Path as _Path

REAL_DIR = _Path(__file__).resolve().parent / "assets" / "config1"
SACRE_COEUR = ["sacre_coeur_A.jpg", "sacre_coeur_B.jpg", "sacre_coeur_B180.jpg", "sacre_coeur_B90.jpg", "sacre_coeur_squared.jpg"]  # sorted: the reference's image list order
PYTEST_IMAGES = ["DSC_6466.jpg", "DSC_6467.jpg", "DSC_6468.jpg"]
CONFIG1_SP = {"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": 2000, "remove_borders": 4, "fix_sampling": False}   # config/superpoint+lightglue.yaml:10-16
CONFIG1_LG = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}                                          # ...yaml:18-24
CONFIG1_AL = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 3}              # config.py:197-204
DESC_STRIDE = 16      # full descriptors are stored for every 16th keypoint, 4 fixed random projections for all of them


def config1_pairs(names=None):
    """pairs_from_bruteforce (pairs_generator.py:37-38): combinations of the sorted image list."""
    from itertools import combinations
    return list(combinations(range(len(names or SACRE_COEUR)), 2))


def real_rgb(name: str):
    """(H, W, 3) uint8, as rasterio's read() transposed to HWC gives it (extractor_base.py:190-196)."""
    import numpy as np
    from PIL import Image
    return np.asarray(Image.open(str(REAL_DIR / name)).convert("RGB"))


def real_gray(name: str):
    """What ExtractorBase.extract hands to SuperPoint's _extract: cv2.cvtColor(RGB array, COLOR_BGR2GRAY) in 8-bit fixed point — quirk Q5,
    the R / B weights end up swapped — then astype(float32), values 0..255 (extractor_base.py:197-202)."""
    import numpy as np
    a = real_rgb(name).astype(np.int64)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(np.uint8).astype(np.float32)


def pixel_digest(arr) -> str:
    import hashlib
    import numpy as np
    a = np.ascontiguousarray(arr)
    return hashlib.sha1(str(a.shape).encode() + a.tobytes()).hexdigest()


def desc_projection(dim: int):
    """(dim, 4) float64 fixed random directions of ~unit length: desc.T @ P is stored for EVERY keypoint."""
    g = torch.Generator().manual_seed(4242 + dim)
    return (torch.randn(dim, 4, generator=g, dtype=torch.float64) / math.sqrt(dim)).numpy()


def fp16_round_trip(feats: dict) -> dict:
    """save_features_h5 (extractor_base.py:56-86, quirk Q6): every float32 array is stored as float16; the matcher reads that back."""
    import numpy as np
    return {k: (np.asarray(v).astype(np.float16).astype(np.float32) if np.asarray(v).dtype == np.float32 else np.asarray(v)) for k, v in feats.items()}


def real_mosaic(height: int, width: int):
    """(height, width, 3) uint8 canvas tiled with the eight real photographs (shelf packing in list order, cropped at the canvas edges, repeated
    until full): a large-format, texture-rich input for the tile-sized tests (config/aliked.yaml: 2000 x 2000 tiles, 8000 keypoints)."""
    import numpy as np
    canvas = np.zeros((height, width, 3), np.uint8)
    photos = [real_rgb(n) for n in SACRE_COEUR + PYTEST_IMAGES]
    y, i = 0, 0
    while y < height:
        x, shelf = 0, 0
        while x < width:
            p = photos[i % len(photos)]
            i += 1
            h, w = min(p.shape[0], height - y), min(p.shape[1], width - x)
            canvas[y:y + h, x:x + w] = p[:h, :w]
            x += p.shape[1]
            shelf = max(shelf, p.shape[0])
        y += shelf
    return canvas
