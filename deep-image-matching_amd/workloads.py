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
