"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the
SuperPoint forward that deep-image-matching runs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  It is a functional torch-CPU restatement of the
reference network file
  SPN = src/deep_image_matching/thirdparty/SuperGluePretrainedNetwork/models/superpoint.py
written stage by stage so every intermediate tensor is a parity tap.  It is pinned
against the reference module itself by oracle/make_golden.py (run in the build
container, where /root/reference exists) and against the committed vectors in
tests/golden/ by tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

DEFAULT_CFG = {  # SPN:112-118
    "nms_radius": 4,
    "keypoint_threshold": 0.005,
    "max_keypoints": -1,
    "remove_borders": 4,
    "fix_sampling": False,  # extractors/superpoint.py:16-27,56-57 (Q3)
}


def _conv(x, sd, name, relu=True, pad=1):
    y = F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=pad)
    return F.relu(y) if relu else y


def encoder(image: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """SPN:161-171 — eight 3x3 conv+ReLU, a 2x2/2 max-pool after each of the first three pairs."""
    x = image
    for i, (a, b) in enumerate((("conv1a", "conv1b"), ("conv2a", "conv2b"), ("conv3a", "conv3b"), ("conv4a", "conv4b"))):
        x = _conv(_conv(x, sd, a), sd, b)
        if i < 3:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
    return x


def detector_logits(x: torch.Tensor, sd) -> torch.Tensor:
    """SPN:174-175 — convPa(3x3)+ReLU, convPb(1x1): [B,65,h,w]."""
    return _conv(_conv(x, sd, "convPa"), sd, "convPb", relu=False, pad=0)


def score_map(logits: torch.Tensor) -> torch.Tensor:
    """SPN:176-179 — softmax over the 65 channels, drop the dustbin, 8x8 depth-to-space
    (channel c of cell (i,j) lands at pixel (8i + c//8, 8j + c%8))."""
    p = torch.softmax(logits, dim=1)[:, :64]
    b, _, h, w = p.shape
    p = p.reshape(b, 8, 8, h, w)  # [b, dy, dx, i, j]
    return p.permute(0, 3, 1, 4, 2).reshape(b, h * 8, w * 8)


def simple_nms(scores: torch.Tensor, radius: int) -> torch.Tensor:
    """SPN:47-63 — max-mask, then two rounds of suppress-and-recover.  P(.) is a
    (2r+1)^2 stride-1 max-pool with -inf padding; ties are all kept (exact ==)."""
    assert radius >= 0
    k = 2 * radius + 1

    def P(t):
        return F.max_pool2d(t, kernel_size=k, stride=1, padding=radius)

    keep = scores == P(scores)
    for _ in range(2):
        near_kept = P(keep.to(scores.dtype)) > 0
        rest = scores.masked_fill(near_kept, 0.0)
        keep = keep | ((rest == P(rest)) & ~near_kept)
    return torch.where(keep, scores, torch.zeros_like(scores))


def select_keypoints(nms: torch.Tensor, threshold: float, border: int, max_keypoints: int):
    """SPN:183-207 for ONE image [H8, W8]: row-major nonzero(s > thr) as (y, x), border
    filter against the score-map size, then top-k (score-descending) only when more than
    k survive.  Returns (yx int64 [N,2], scores [N])."""
    H8, W8 = nms.shape
    yx = torch.nonzero(nms > threshold)
    sc = nms[yx[:, 0], yx[:, 1]]
    ok = (yx[:, 0] >= border) & (yx[:, 0] < H8 - border) & (yx[:, 1] >= border) & (yx[:, 1] < W8 - border)
    yx, sc = yx[ok], sc[ok]
    if max_keypoints >= 0 and max_keypoints < len(yx):
        sc, idx = torch.topk(sc, max_keypoints, dim=0)
        yx = yx[idx]
    return yx, sc


def dense_descriptors(x: torch.Tensor, sd, normalize: bool = True) -> torch.Tensor:
    """SPN:213-215 — convDa(3x3)+ReLU, convDb(1x1), L2-normalise over channels."""
    d = _conv(_conv(x, sd, "convDa"), sd, "convDb", relu=False, pad=0)
    return F.normalize(d, p=2, dim=1) if normalize else d


def sample_descriptors(kpts_xy: torch.Tensor, desc: torch.Tensor, fix_sampling: bool = False, s: int = 8):
    """SPN:81-98 (original: centre-of-cell grid, align_corners=True) or the DIM/hloc
    'fix_sampling' variant (extractors/superpoint.py:16-27: (k+0.5)/(w*s), align_corners=False).
    kpts_xy [N,2] (x,y) float32; desc [1,C,h,w] L2-normalised.  Returns [C,N]."""
    _, c, h, w = desc.shape
    k = kpts_xy.clone().to(torch.float32)
    if fix_sampling:
        k = (k + 0.5) / (k.new_tensor([w, h]) * s)
        k = k * 2 - 1
        ac = False
    else:
        k = k - s / 2 + 0.5
        k = k / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(k)[None]
        k = k * 2 - 1
        ac = True
    out = F.grid_sample(desc, k.view(1, 1, -1, 2), mode="bilinear", align_corners=ac)
    return F.normalize(out.reshape(1, c, -1), p=2, dim=1)[0]


@torch.no_grad()
def superpoint_forward(image: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: Optional[dict] = None, taps: bool = False):
    """image [1,1,H,W] float32 in [0,1].  Returns the reference's output dict for one
    image: keypoints (N,2) float32 (x,y), scores (N,), descriptors (256,N); with
    taps=True also the intermediate tensors."""
    cfg = {**DEFAULT_CFG, **(cfg or {})}
    assert image.dim() == 4 and image.shape[0] == 1 and image.shape[1] == 1
    x = encoder(image, sd)
    logits = detector_logits(x, sd)
    smap = score_map(logits)
    nms = simple_nms(smap, cfg["nms_radius"])
    yx, sc = select_keypoints(nms[0], cfg["keypoint_threshold"], cfg["remove_borders"], cfg["max_keypoints"])
    kpts = torch.flip(yx, [1]).float()  # SPN:210
    dense = dense_descriptors(x, sd)
    desc = sample_descriptors(kpts, dense, cfg["fix_sampling"])
    out = {"keypoints": kpts, "scores": sc, "descriptors": desc}
    if taps:
        out.update(encoder=x, logits=logits, score_map=smap, nms_map=nms, dense_desc=dense)
    return out
