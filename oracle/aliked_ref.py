"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the ALIKED
forward that deep-image-matching runs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Functional torch-CPU restatement of
  ALN = src/deep_image_matching/thirdparty/LightGlue/lightglue/aliked.py
as driven by extractors/aliked.py:45-64, INCLUDING the reference's quirks:
  Q7  the plugin never calls .eval(), so every BatchNorm2d normalises with the CURRENT batch
      statistics (biased variance over N*H*W); running stats are ignored (ALX:40-43);
  Q8  ALIKED.forward unpacks DKD's (keypoints, dispersity, scores) as (keypoints, scores,
      dispersity), so the exported "keypoint_scores" are the score DISPERSITIES (ALN:244,682,692).
`deform_conv2d` restates torchvision.ops.deform_conv2d (v0.22, not vendored in the reference;
call site ALN:322-329): offsets ordered (dy, dx) per tap, zero outside, no mask.  Pinned by
oracle/make_golden.py against the reference file itself (imported with torchvision/kornia
stubbed, this deform_conv2d standing in for the absent torchvision op) — so the network wiring
is pinned by the reference, the deformable sampling rule only by torchvision's documentation:
"parity unpinned" for that one op.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

CFGS = {  # ALN:573-579  c1, c2, c3, c4, dim, K, M
    "aliked-t16": (8, 16, 32, 64, 64, 3, 16),
    "aliked-n16": (16, 32, 64, 128, 128, 3, 16),
    "aliked-n16rot": (16, 32, 64, 128, 128, 3, 16),
    "aliked-n32": (16, 32, 64, 128, 128, 3, 32),
}
DEFAULT_CFG = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 2}
N_LIMIT_MAX = 20000  # ALN:571


def _bilinear_zero(x: torch.Tensor, py: torch.Tensor, px: torch.Tensor) -> torch.Tensor:
    """torchvision deform_conv2d's bilinear_interpolate: x [B,C,H,W], py/px [B,Ho,Wo] ->
    [B,C,Ho,Wo]; 0 when the point is outside (-1, H) x (-1, W); missing corners contribute 0."""
    B, C, H, W = x.shape
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    hy, hx = 1 - ly, 1 - lx
    y0, x0 = y0.long(), x0.long()
    y1, x1 = y0 + 1, x0 + 1
    flat = x.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *yy.shape[1:])
        return v * ok[:, None].to(x.dtype)

    out = (hy * hx)[:, None] * tap(y0, x0) + (hy * lx)[:, None] * tap(y0, x1) + (ly * hx)[:, None] * tap(y1, x0) + (ly * lx)[:, None] * tap(y1, x1)
    return out * inside[:, None].to(x.dtype)


def deform_conv2d(x: torch.Tensor, offset: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, padding: int = 1):
    """3x3 / stride 1 deformable convolution, offset [B, 2*K*K, H, W] = (dy, dx) per tap (k = ky*K + kx)."""
    B, C, H, W = x.shape
    Co, _, K, _ = weight.shape
    ys = torch.arange(H, dtype=x.dtype)[None, :, None]
    xs = torch.arange(W, dtype=x.dtype)[None, None, :]
    cols = []
    for k in range(K * K):
        ky, kx = divmod(k, K)
        cols.append(_bilinear_zero(x, ys - padding + ky + offset[:, 2 * k], xs - padding + kx + offset[:, 2 * k + 1]))
    col = torch.stack(cols, 2)  # [B, C, KK, H, W]
    out = torch.einsum("bckhw,ock->bohw", col, weight.reshape(Co, C, K * K))
    return out if bias is None else out + bias[None, :, None, None]


def _bn_train(x, sd, name):
    """BatchNorm2d in TRAINING mode (Q7): batch mean, biased batch variance, eps 1e-5, affine."""
    return F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], True, 0.1, 1e-5)


def _conv(x, sd, name, conv_type):
    """get_conv (ALN:332-363): plain 3x3 (no bias) or DeformableConv2d (ALN:274-330)."""
    if conv_type == "conv":
        return F.conv2d(x, sd[name + ".weight"], None, padding=1)
    h, w = x.shape[2:]
    max_offset = max(h, w) / 4.0
    off = F.conv2d(x, sd[name + ".offset_conv.weight"], sd[name + ".offset_conv.bias"], padding=1).clamp(-max_offset, max_offset)
    return deform_conv2d(x, off, sd[name + ".regular_conv.weight"], None, padding=1)


def conv_block(x, sd, p, conv_type="conv"):
    """ConvBlock (ALN:367-393)."""
    x = F.selu(_bn_train(_conv(x, sd, p + ".conv1", conv_type), sd, p + ".bn1"))
    return F.selu(_bn_train(_conv(x, sd, p + ".conv2", conv_type), sd, p + ".bn2"))


def res_block(x, sd, p, conv_type):
    """ResBlock (ALN:396-450) with the 1x1 (biased) downsample of get_resblock (ALN:631-642)."""
    out = F.selu(_bn_train(_conv(x, sd, p + ".conv1", conv_type), sd, p + ".bn1"))
    out = _bn_train(_conv(out, sd, p + ".conv2", conv_type), sd, p + ".bn2")
    identity = F.conv2d(x, sd[p + ".downsample.weight"], sd[p + ".downsample.bias"])
    return F.selu(out + identity)


def dense_maps(image: torch.Tensor, sd: Dict[str, torch.Tensor]):
    """extract_dense_map (ALN:644-675): returns (feature_map [1,dim,H,W] L2-normalised, score_map [1,1,H,W])."""
    H, W = image.shape[-2:]
    ph, pw = ((H // 32 + 1) * 32 - H) % 32, ((W // 32 + 1) * 32 - W) % 32  # InputPadder (ALN:247-271)
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    x = F.pad(image, pad, mode="replicate")
    x1 = conv_block(x, sd, "block1", "conv")
    x2 = res_block(F.avg_pool2d(x1, 2, 2), sd, "block2", "conv")
    x3 = res_block(F.avg_pool2d(x2, 4, 4), sd, "block3", "dcn")
    x4 = res_block(F.avg_pool2d(x3, 4, 4), sd, "block4", "dcn")
    f1 = F.selu(F.conv2d(x1, sd["conv1.weight"]))
    f2 = F.selu(F.conv2d(x2, sd["conv2.weight"]))
    f3 = F.selu(F.conv2d(x3, sd["conv3.weight"]))
    f4 = F.selu(F.conv2d(x4, sd["conv4.weight"]))
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=True)  # noqa: E731
    x1234 = torch.cat([f1, up(f2, 2), up(f3, 8), up(f4, 32)], dim=1)
    s = F.selu(F.conv2d(x1234, sd["score_head.0.weight"]))
    s = F.selu(F.conv2d(s, sd["score_head.2.weight"], padding=1))
    s = F.selu(F.conv2d(s, sd["score_head.4.weight"], padding=1))
    score = torch.sigmoid(F.conv2d(s, sd["score_head.6.weight"], padding=1))
    feat = F.normalize(x1234, p=2, dim=1)
    hs, ws = slice(pad[2], feat.shape[-2] - pad[3]), slice(pad[0], feat.shape[-1] - pad[1])
    return feat[..., hs, ws], score[..., hs, ws]


def _simple_nms(scores, r):
    """ALN:66-89 (identical to SuperPoint's simple_nms)."""
    k = 2 * r + 1
    P = lambda t: F.max_pool2d(t, kernel_size=k, stride=1, padding=r)  # noqa: E731
    keep = scores == P(scores)
    for _ in range(2):
        near = P(keep.to(scores.dtype)) > 0
        rest = scores.masked_fill(near, 0.0)
        keep = keep | ((rest == P(rest)) & ~near)
    return torch.where(keep, scores, torch.zeros_like(scores))


def dkd_nms_map(score_map: torch.Tensor, radius: int) -> torch.Tensor:
    """simple_nms + the border clearing of DKD.forward (ALN:136-147, image_size=None)."""
    _, _, h, w = score_map.shape
    nms = _simple_nms(score_map, radius)
    nms[:, :, :radius, :] = 0
    nms[:, :, :, :radius] = 0
    nms[:, :, h - radius:, :] = 0
    nms[:, :, :, w - radius:] = 0
    return nms


def dkd_select(score_map: torch.Tensor, radius: int, scores_th: float, n_limit: int, top_k: int = -1) -> torch.Tensor:
    """The flat pixel indices DKD.forward detects (ALN:149-174)."""
    nms = dkd_nms_map(score_map, radius)
    flat = score_map.reshape(-1)
    if top_k > 0:
        return torch.topk(nms.reshape(-1), top_k).indices
    if scores_th > 0:
        mask = nms > scores_th
        if mask.sum() == 0:
            mask = nms > flat.mean()
    else:
        mask = nms > flat.mean()
    idx = mask.reshape(-1).nonzero()[:, 0]
    if len(idx) > n_limit:
        order = flat[idx].sort(descending=True)[1]
        idx = idx[order[:n_limit]]
    return idx


def dkd_refine(score_map: torch.Tensor, idx: torch.Tensor, radius: int):
    """The sub-pixel half of DKD.forward at the given flat pixel indices (ALN:176-216)."""
    _, _, h, w = score_map.shape
    ks = 2 * radius + 1
    patches = F.unfold(score_map, kernel_size=ks, padding=radius)[0].t()[idx]  # [M, ks*ks], zero padded
    g = torch.linspace(-radius, radius, ks, dtype=score_map.dtype)   # fp32 in the reference; the fp64 yardstick runs pass doubles
    gy, gx = torch.meshgrid(g, g, indexing="ij")
    grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], 1)  # (x, y) offsets, row-major over the patch
    xy = torch.stack([idx % w, torch.div(idx, w, rounding_mode="trunc")], 1)
    mx = patches.max(dim=1).values[:, None]
    e = ((patches - mx) / 0.1).exp()
    res = e @ grid / e.sum(1)[:, None]
    d2 = (torch.norm((grid[None] - res[:, None]) / radius, dim=-1) ** 2)
    disp = (e * d2).sum(1) / e.sum(1)
    wh = torch.tensor([w - 1, h - 1], dtype=score_map.dtype)
    kp = (xy + res) / wh * 2 - 1
    ksc = F.grid_sample(score_map, kp.view(1, 1, -1, 2), mode="bilinear", align_corners=True)[0, 0, 0]
    return kp, disp, ksc


def dkd(score_map: torch.Tensor, radius: int, scores_th: float, n_limit: int, top_k: int = -1, idx: Optional[torch.Tensor] = None):
    """DKD.forward for one image (ALN:123-244, sub_pixel=True).  Returns keypoints in [-1,1]
    (x,y), score dispersity, bilinear keypoint score.  ``idx``: evaluate at these flat pixel indices instead of DKD's own detections
    (tests of the top-k mode's zero-score fill, whose pixels torch.topk picks by the accidents of its sort)."""
    if idx is None:
        idx = dkd_select(score_map, radius, scores_th, n_limit, top_k)
    return dkd_refine(score_map, idx, radius)


def _get_patches(feat: torch.Tensor, corners_xy: torch.Tensor, ps: int) -> torch.Tensor:
    """get_patches (ALN:48-63): feat [C,H,W], integer (x,y) -> [N,C,ps,ps]."""
    c, h, w = feat.shape
    corner = (corners_xy - ps / 2 + 1).long()
    cx = corner[:, 0].clamp(min=0, max=w - 1 - ps)
    cy = corner[:, 1].clamp(min=0, max=h - 1 - ps)
    off = torch.arange(ps)
    yy = cy[:, None, None] + off[None, :, None]  # [N, ps, 1]
    xx = cx[:, None, None] + off[None, None, :]  # [N, 1, ps]
    return feat[:, yy, xx].permute(1, 0, 2, 3)  # [N, C, ps(y), ps(x)]


def sddh(feat: torch.Tensor, kpts: torch.Tensor, sd, K: int, M: int) -> torch.Tensor:
    """SDDH.forward for one image (ALN:503-558): feat [1,C,H,W], kpts [N,2] in [-1,1] -> [N,C]."""
    _, c, h, w = feat.shape
    wh = torch.tensor([[w - 1, h - 1]], dtype=feat.dtype)
    max_offset = max(h, w) / 4.0
    kwh = (kpts / 2 + 0.5) * wh
    n = kpts.shape[0]
    patch = _get_patches(feat[0], kwh.long(), K)
    off = F.conv2d(patch, sd["desc_head.offset_conv.0.weight"], sd["desc_head.offset_conv.0.bias"])
    off = F.conv2d(F.selu(off), sd["desc_head.offset_conv.2.weight"], sd["desc_head.offset_conv.2.bias"]).clamp(-max_offset, max_offset)
    off = off[:, :, 0, 0].view(n, 2, M).permute(0, 2, 1)  # [N, M, 2]
    pos = 2.0 * (kwh.unsqueeze(1) + off) / wh[None] - 1
    f = F.grid_sample(feat, pos.reshape(1, n * M, 1, 2), mode="bilinear", align_corners=True)
    f = f.reshape(c, n, M, 1).permute(1, 0, 2, 3)  # [N, C, M, 1]
    f = F.selu(F.conv2d(f, sd["desc_head.sf_conv.weight"])).squeeze(-1)  # [N, C, M]
    d = torch.einsum("ncp,pcd->nd", f, sd["desc_head.agg_weights"])
    return F.normalize(d, p=2.0, dim=1)


@torch.no_grad()
def aliked_forward(image: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: Optional[dict] = None, taps: bool = False,
                   idx: Optional[torch.Tensor] = None):
    """image [1,3,H,W] (or [1,1,H,W]) float32 in [0,1].  Returns DIM's feature dict for one image:
    keypoints (N,2) pixel (x,y), descriptors (D,N), scores (N,) (= dispersities, Q8)."""
    cfg = {**DEFAULT_CFG, **(cfg or {})}
    c1, c2, c3, c4, dim, K, M = CFGS[cfg["model_name"]]
    if image.shape[1] == 1:
        image = image.repeat(1, 3, 1, 1)  # kornia grayscale_to_rgb
    feat, score = dense_maps(image, sd)
    th = cfg["detection_threshold"]
    mk = cfg["max_num_keypoints"]
    kp, disp, ksc = dkd(score, cfg["nms_radius"], th, mk if mk > 0 else N_LIMIT_MAX, top_k=-1 if th > 0 else mk, idx=idx)
    desc = sddh(feat, kp, sd, K, M)
    h, w = image.shape[-2:]
    wh = torch.tensor([w - 1, h - 1], dtype=image.dtype)
    out = {"keypoints": wh * (kp + 1) / 2.0, "descriptors": desc.t().contiguous(), "scores": disp}
    if taps:
        out.update(feature_map=feat, score_map=score, keypoint_scores_true=ksc, kpts_norm=kp)
    return out
