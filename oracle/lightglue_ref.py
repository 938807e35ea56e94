"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the
LightGlue forward on the path deep-image-matching takes on CPU.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  Functional torch-CPU restatement of
  LGN = src/deep_image_matching/thirdparty/LightGlue/lightglue/lightglue.py
for batch size 1 (DIM always runs B=1: matchers/lightglue.py:58), CPU branch
(fp32 attention, ONE cross similarity with row- and column-softmax LGN:197-206,
pruning threshold -1 i.e. always on when width_confidence > 0, LGN:318-323 / Q9).
Pinned against the reference module by oracle/make_golden.py and against
tests/golden/ by tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

DEFAULT_CONF = {  # LGN:301-314
    "n_layers": 9,
    "num_heads": 4,
    "depth_confidence": 0.95,
    "width_confidence": 0.99,
    "filter_threshold": 0.1,
}


def normalize_keypoints(kpts: torch.Tensor, size: torch.Tensor) -> torch.Tensor:
    """LGN:25-34 — (k - size/2) / (max(size)/2); `size` is used as given ((H,W) from DIM, Q4)."""
    size = size.to(kpts)
    return (kpts - (size / 2)[None, :]) / (size.max() / 2)


def positional_encoding(kpts_n: torch.Tensor, Wr: torch.Tensor) -> torch.Tensor:
    """LGN:57-70 — Wr (32x2, no bias) -> cos, sin, each value repeated twice along the
    last axis.  Returns [2, N, 64]."""
    proj = kpts_n @ Wr.t()
    return torch.stack([torch.cos(proj), torch.sin(proj)], 0).repeat_interleave(2, dim=-1)


def _rot_half(t: torch.Tensor) -> torch.Tensor:
    """LGN:41-44 — pairs (t0, t1) -> (-t1, t0)."""
    a, b = t[..., 0::2], t[..., 1::2]
    return torch.stack((-b, a), dim=-1).flatten(-2)


def _rotary(enc: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """LGN:47-54."""
    return t * enc[0] + _rot_half(t) * enc[1]


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ffn(x, msg, sd, prefix):
    """LGN:139-144,159 — Linear(512,512), LayerNorm(512), GELU(erf), Linear(512,256) on cat[x, msg]."""
    h = _lin(torch.cat([x, msg], -1), sd, prefix + ".ffn.0")
    h = F.layer_norm(h, (h.shape[-1],), sd[prefix + ".ffn.1.weight"], sd[prefix + ".ffn.1.bias"], 1e-5)
    return _lin(F.gelu(h), sd, prefix + ".ffn.3")


def self_block(x: torch.Tensor, enc: torch.Tensor, sd, i: int, heads: int = 4) -> torch.Tensor:
    """LGN:146-159 — x [N,256], enc [2,N,64]."""
    p = f"transformers.{i}.self_attn"
    n, d = x.shape
    qkv = _lin(x, sd, p + ".Wqkv").reshape(n, heads, d // heads, 3).permute(1, 0, 2, 3)  # [h, N, 64, 3]
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q, k = _rotary(enc[:, None], q), _rotary(enc[:, None], k)
    if n == 0:
        ctx = q.new_zeros((heads, n, d // heads))
    else:
        attn = torch.softmax((q @ k.transpose(-1, -2)) * (d // heads) ** -0.5, dim=-1)
        ctx = attn @ v
    msg = _lin(ctx.permute(1, 0, 2).reshape(n, d), sd, p + ".out_proj")
    return x + _ffn(x, msg, sd, p)


def cross_block(x0: torch.Tensor, x1: torch.Tensor, sd, i: int, heads: int = 4):
    """LGN:186-211, CPU branch :197-206."""
    p = f"transformers.{i}.cross_attn"
    d = x0.shape[-1]
    dh = d // heads

    def split(t):
        return t.reshape(t.shape[0], heads, dh).permute(1, 0, 2)

    qk0, qk1 = split(_lin(x0, sd, p + ".to_qk")), split(_lin(x1, sd, p + ".to_qk"))
    v0, v1 = split(_lin(x0, sd, p + ".to_v")), split(_lin(x1, sd, p + ".to_v"))
    s = (dh ** -0.5) ** 0.5
    sim = (qk0 * s) @ (qk1 * s).transpose(-1, -2)  # [h, M, N]
    m0 = torch.softmax(sim, dim=-1) @ v1
    m1 = torch.softmax(sim.transpose(-1, -2).contiguous(), dim=-1) @ v0
    m0 = _lin(m0.permute(1, 0, 2).reshape(-1, d), sd, p + ".to_out")
    m1 = _lin(m1.permute(1, 0, 2).reshape(-1, d), sd, p + ".to_out")
    return x0 + _ffn(x0, m0, sd, p), x1 + _ffn(x1, m1, sd, p)


def log_assignment(d0: torch.Tensor, d1: torch.Tensor, sd, i: int) -> torch.Tensor:
    """LGN:246-275 — returns the (M+1)x(N+1) log assignment matrix."""
    p = f"log_assignment.{i}"
    dim = d0.shape[-1]
    md0, md1 = _lin(d0, sd, p + ".final_proj") / dim ** 0.25, _lin(d1, sd, p + ".final_proj") / dim ** 0.25
    sim = md0 @ md1.t()
    z0, z1 = _lin(d0, sd, p + ".matchability"), _lin(d1, sd, p + ".matchability")  # [M,1], [N,1]
    m, n = sim.shape
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).t()
    s0 = F.log_softmax(sim, 1)
    s1 = F.log_softmax(sim.t().contiguous(), 1).t()
    scores = sim.new_zeros((m + 1, n + 1))
    scores[:m, :n] = s0 + s1 + cert
    scores[:-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[-1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def filter_matches(scores: torch.Tensor, th: float):
    """LGN:281-297 on one [M+1, N+1] matrix (first-max tie rule of Tensor.max on CPU)."""
    inner = scores[:-1, :-1]
    max0, max1 = inner.max(1), inner.max(0)
    m0, m1 = max0.indices, max1.indices
    mutual0 = torch.arange(m0.shape[0]) == m1[m0]
    mutual1 = torch.arange(m1.shape[0]) == m0[m1]
    e0 = max0.values.exp()
    ms0 = torch.where(mutual0, e0, e0.new_tensor(0))
    ms1 = torch.where(mutual1, ms0[m1], e0.new_tensor(0))
    valid0 = mutual0 & (ms0 > th)
    valid1 = mutual1 & valid0[m1]
    return torch.where(valid0, m0, -1), torch.where(valid1, m1, -1), ms0, ms1


@torch.no_grad()
def lightglue_forward(kpts0, desc0, size0, kpts1, desc1, size1, sd: Dict[str, torch.Tensor],
                      conf: Optional[dict] = None, taps: bool = False):
    """kptsX [N,2] float32 pixel (x,y); descX [N,D]; sizeX tensor([a,b]) as fed by DIM
    (image_size, (H,W), Q4).  Returns the reference's outputs for B=1 with the batch
    dimension dropped."""
    c = {**DEFAULT_CONF, **(conf or {})}
    L, heads = c["n_layers"], c["num_heads"]
    m, n = kpts0.shape[0], kpts1.shape[0]
    # conf["dtype"] = torch.float64 (with a float64 state dict) evaluates the same network in double precision: the
    # yardstick the fp32 paths are measured against (tests/test_saturation_gpu.py); the default is the reference's fp32
    dt = c.get("dtype", torch.float32)
    k0 = normalize_keypoints(kpts0.to(dt), size0)
    k1 = normalize_keypoints(kpts1.to(dt), size1)
    d0, d1 = desc0.to(dt).contiguous(), desc1.to(dt).contiguous()
    if "input_proj.weight" in sd:  # LGN:361-364,473-474
        d0, d1 = _lin(d0, sd, "input_proj"), _lin(d1, sd, "input_proj")
    e0 = positional_encoding(k0, sd["posenc.Wr.weight"])
    e1 = positional_encoding(k1, sd["posenc.Wr.weight"])
    thr = sd["confidence_thresholds"]

    early = c["depth_confidence"] > 0
    prune = c["width_confidence"] > 0
    ind0, ind1 = torch.arange(m), torch.arange(n)
    prune0, prune1 = torch.ones(m, dtype=torch.long), torch.ones(n, dtype=torch.long)
    tok0 = tok1 = None
    layer_taps = []
    i = 0
    for i in range(L):
        if d0.shape[0] == 0 or d1.shape[0] == 0:
            break
        d0 = self_block(d0, e0, sd, i, heads)
        d1 = self_block(d1, e1, sd, i, heads)
        d0, d1 = cross_block(d0, d1, sd, i, heads)
        if taps:
            layer_taps.append((d0.clone(), d1.clone(), ind0.clone(), ind1.clone()))
        if i == L - 1:
            continue
        if early:  # LGN:497-500, 593-604
            tn = f"token_confidence.{i}.token.0"
            tok0 = torch.sigmoid(_lin(d0, sd, tn)).squeeze(-1)
            tok1 = torch.sigmoid(_lin(d1, sd, tn)).squeeze(-1)
            conf_all = torch.cat([tok0, tok1], -1)
            ratio = 1.0 - (conf_all < thr[i]).float().sum() / (m + n)
            if ratio > c["depth_confidence"]:
                break
        if prune:  # LGN:501-516, 586-591 (CPU: threshold -1 -> always)
            mn = f"log_assignment.{i}.matchability"
            for side in (0, 1):
                d, e, ind, pr, tok = (d0, e0, ind0, prune0, tok0) if side == 0 else (d1, e1, ind1, prune1, tok1)
                keep = torch.sigmoid(_lin(d, sd, mn)).squeeze(-1) > (1 - c["width_confidence"])
                if tok is not None:
                    keep = keep | (tok <= thr[i])
                kidx = torch.where(keep)[0]
                ind = ind[kidx]
                d = d[kidx]
                e = e[:, kidx]
                pr[ind] += 1
                if side == 0:
                    d0, e0, ind0 = d, e, ind
                else:
                    d1, e1, ind1 = d, e, ind

    if d0.shape[0] == 0 or d1.shape[0] == 0:  # LGN:518-540
        out = {
            "matches0": torch.full((m,), -1, dtype=torch.long), "matches1": torch.full((n,), -1, dtype=torch.long),
            "matching_scores0": torch.zeros(m), "matching_scores1": torch.zeros(n), "stop": i + 1,
            "matches": torch.empty((0, 2), dtype=torch.long), "scores": torch.empty((0,)),
            "prune0": prune0 if prune else torch.ones(m) * L, "prune1": prune1 if prune else torch.ones(n) * L,
        }
        return out

    scores = log_assignment(d0, d1, sd, i)
    a0, a1, ms0, ms1 = filter_matches(scores, c["filter_threshold"])
    valid = a0 > -1
    mi0 = torch.where(valid)[0]
    mi1 = a0[valid]
    matches = torch.stack([ind0[mi0], ind1[mi1]], -1) if prune else torch.stack([mi0, mi1], -1)
    mscores = ms0[valid]
    if prune:  # LGN:556-566 scatter back to the un-pruned index space
        f0 = torch.full((m,), -1, dtype=torch.long)
        f1 = torch.full((n,), -1, dtype=torch.long)
        f0[ind0] = torch.where(a0 == -1, -1, ind1[a0.clamp(min=0)])
        f1[ind1] = torch.where(a1 == -1, -1, ind0[a1.clamp(min=0)])
        s0 = torch.zeros(m, dtype=ms0.dtype)
        s1 = torch.zeros(n, dtype=ms0.dtype)
        s0[ind0] = ms0
        s1[ind1] = ms1
        a0, a1, ms0, ms1 = f0, f1, s0, s1
    else:
        prune0, prune1 = torch.ones(m) * L, torch.ones(n) * L
    out = {
        "matches0": a0, "matches1": a1, "matching_scores0": ms0, "matching_scores1": ms1, "stop": i + 1,
        "matches": matches, "scores": mscores, "prune0": prune0, "prune1": prune1,
    }
    if taps:
        out.update(log_assignment=scores, layers=layer_taps, ind0=ind0, ind1=ind1, enc0=e0, enc1=e1)
    return out
