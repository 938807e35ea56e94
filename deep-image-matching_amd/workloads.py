"""Synthetic workloads with TRUE correspondences for the benchmarks and the parity tests.

The headline benchmark uses independent random images (BASELINE configs[2] asks for nothing else); its match lists are
nearly empty, which is fine for the kernels (LightGlue's work is fixed) but leaves geometric verification, the match writers and
the multi-GPU match gather without work.  ``shifted_crops`` cuts every image of a job out of ONE random canvas at offsets that are
multiples of 8 pixels: SuperPoint is equivariant under such shifts (three 2 x 2 pools, 8 x 8 cells; away from the borders the
features of a scene point are identical in every crop), so any two images share hundreds of keypoints with equal descriptors and
``weights.synthetic_lightglue_matching_state_dict`` matches them; the true relative geometry of a pair is a pure translation.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shifted_crops(n_images: int, H: int, W: int, max_shift: int = 256, seed: int = 0, canvas: str = "noise") -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (images [n, H, W] float32 in [0, 1], offsets [n, 2] int64 (dy, dx), multiples of 8 in [0, max_shift])."""
    g = torch.Generator().manual_seed(seed)
    steps = max_shift // 8 + 1
    off = torch.randint(0, steps, (n_images, 2), generator=g) * 8
    off[0] = 0
    ch, cw = H + max_shift, W + max_shift
    if canvas == "noise":
        base = torch.rand(ch, cw, generator=g)
    else:   # smooth blobs + noise: fewer, stronger maxima
        yy, xx = torch.meshgrid(torch.arange(ch, dtype=torch.float32), torch.arange(cw, dtype=torch.float32), indexing="ij")
        base = torch.zeros(ch, cw)
        for _ in range(200):
            cy, cx = torch.rand(2, generator=g) * torch.tensor([ch, cw], dtype=torch.float32)
            sdev = 2.0 + 6.0 * torch.rand(1, generator=g)
            base += torch.rand(1, generator=g) * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sdev * sdev))
        base = (base / base.max() * 0.9 + 0.05 * torch.rand(ch, cw, generator=g)).clamp(0, 1)
    imgs = torch.stack([base[int(dy):int(dy) + H, int(dx):int(dx) + W] for dy, dx in off.tolist()]).contiguous()
    return imgs, off


def true_match_fraction(kp0: torch.Tensor, kp1: torch.Tensor, matches: torch.Tensor, off0, off1, tol: float = 1.5) -> float:
    """Fraction of ``matches`` (S, 2) whose keypoints are the same canvas point (pure translation off0 -> off1, (dy, dx))."""
    if matches.numel() == 0:
        return 0.0
    d = kp0[matches[:, 0]] - kp1[matches[:, 1]]
    want = torch.tensor([float(off1[1] - off0[1]), float(off1[0] - off0[0])])
    return float(((d - want).abs().max(1).values < tol).float().mean())


@torch.no_grad()
def descriptor_mean(extractor, images: torch.Tensor, max_images: int = 8) -> torch.Tensor:
    """Mean descriptor over the keypoints of the first images of a workload (the ``center`` argument of
    weights.synthetic_lightglue_matching_state_dict), from the resident extractor (SuperPointHIP / AlikedHIP ``extract_batch``)."""
    b = min(int(images.shape[0]), int(extractor.max_batch), max_images)
    kp, sc, de, n = extractor.extract_batch(images[:b].contiguous())
    rows = torch.cat([de[i, : int(n[i])] for i in range(b)])
    return rows.mean(0).cpu()


def adaptive_lightglue_workload(n_pairs: int, n_kpts: int = 2048, dim: int = 256, seed: int = 11, lone_fraction: float = 0.25,
                                stops=(3, 4, 5, 6, 7, 8, 9), n_layers: int = 9):
    """A LightGlue batch on which the reference's adaptive depth AND width (LGN:494-516, 586-604) really adapt, under ONE set of weights:

    * matching-capable weights (weights.synthetic_lightglue_matching_state_dict, sharpness 40) — image 1 is a shuffled, jittered copy of image 0
      with graded descriptor noise (the flip-rate study's recipe) and ``lone_fraction`` of its keypoints replaced by unrelated ones;
    * two descriptor coordinates are reserved as "difficulty" channels (the similarity heads ignore them): coordinate dim-2 carries a per-PAIR
      value c_p that the token-confidence heads read (layer i: logit = 8 x + logit(thr_i) + i + 0.5), so that every keypoint of pair p turns
      confident exactly at layer ``stops[p % len(stops)]`` and the depth criterion (ratio > 0.95) stops the pair there (9 = never); the
      unrelated keypoints carry 0 there (confident from the first layer) and -1 on coordinate dim-1, which the matchability heads read
      (sigmoid(-6) < 0.01): they are pruned after the first layer (width criterion), everything else (+1) stays.

    Returns (state_dict, kpts [2P, N, 2], desc [2P, N, dim], counts [2P] int32, sizes [2P, 2], expected_stop [P]) — items 2p / 2p+1 are pair p."""
    import math
    from . import weights
    g = torch.Generator().manual_seed(seed)
    sd = weights.synthetic_lightglue_matching_state_dict(1, dim, n_layers=n_layers, sharpness=40.0)
    thr = weights.lightglue_confidence_thresholds(n_layers)
    cu, cv = dim - 2, dim - 1
    for i in range(n_layers):
        fp = f"log_assignment.{i}.final_proj.weight"
        sd[fp] = sd[fp].clone()
        sd[fp][:, cu:] = 0.0
        sd[fp][cu:, :] = 0.0
        mw = torch.zeros(1, dim); mw[0, cv] = 6.0
        sd[f"log_assignment.{i}.matchability.weight"], sd[f"log_assignment.{i}.matchability.bias"] = mw, torch.zeros(1)
        if i < n_layers - 1:
            tw = torch.zeros(1, dim); tw[0, cu] = 8.0
            t = float(thr[i])
            sd[f"token_confidence.{i}.token.0.weight"] = tw
            sd[f"token_confidence.{i}.token.0.bias"] = torch.tensor([math.log(t / (1.0 - t)) + i + 0.5])
    kp, de, expect = [], [], []
    for p in range(n_pairs):
        s = int(stops[p % len(stops)])
        expect.append(s)
        k0 = torch.rand(n_kpts, 2, generator=g) * 1024.0
        d0 = torch.nn.functional.normalize(torch.randn(n_kpts, dim, generator=g), dim=-1)
        perm = torch.randperm(n_kpts, generator=g)
        k1 = k0[perm] + torch.randn(n_kpts, 2, generator=g) * 2.0
        sigma = torch.exp(torch.rand(n_kpts, 1, generator=g) * (math.log(6.0) - math.log(0.3)) + math.log(0.3))
        d1 = d0[perm] + sigma * torch.randn(n_kpts, dim, generator=g) / math.sqrt(dim)
        lone1 = torch.rand(n_kpts, generator=g) < lone_fraction
        d1[lone1] = torch.randn(int(lone1.sum()), dim, generator=g)
        d1 = torch.nn.functional.normalize(d1, dim=-1)
        lone0 = torch.zeros(n_kpts, dtype=torch.bool)
        lone0[perm[lone1]] = True                     # their partners in image 0 have nothing to match either
        c = -(s - 1) / 8.0
        for d, lone in ((d0, lone0), (d1, lone1)):
            d[:, cu] = torch.where(lone, torch.zeros(n_kpts), torch.full((n_kpts,), c))
            d[:, cv] = torch.where(lone, -torch.ones(n_kpts), torch.ones(n_kpts))
        kp += [k0, k1]
        de += [d0, d1]
    counts = torch.full((2 * n_pairs,), n_kpts, dtype=torch.int32)
    sizes = torch.full((2 * n_pairs, 2), 1024.0)
    return sd, torch.stack(kp).contiguous(), torch.stack(de).contiguous(), counts, sizes, torch.tensor(expect)
