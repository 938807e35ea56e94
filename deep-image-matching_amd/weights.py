"""Weights for the SuperPoint / LightGlue hot path.

The official checkpoints (``superpoint_v1.pth``, ``superpoint_lightglue.pth``) are
downloaded by the reference at run time (SPN:110,147-150; LGN:327-328,381-396) and
are not available offline, so this module provides

* ``load_superpoint_state_dict`` / ``load_lightglue_state_dict`` — accept the official
  files unchanged when a path is given (incl. LightGlue's legacy key rename,
  LGN:389-396), and
* seeded synthetic state dicts in exactly the official key/shape layout
   for tests, smoke and the synthetic benchmark.  They are
  drawn from ``torch.Generator`` streams so that the same seed gives the same
  tensors here, in the golden-vector generator and on the GPU box.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Dict, Optional

import torch

SP_LAYERS = [
    # name, cout, cin, k
    ("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3), ("convPb", 65, 256, 1),
    ("convDa", 256, 128, 3), ("convDb", 256, 256, 1),
]


def synthetic_superpoint_state_dict(seed: int = 1234) -> Dict[str, torch.Tensor]:
    """He-normal conv weights, small biases."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, co, ci, k in SP_LAYERS:
        fan_in = ci * k * k
        sd[f"{name}.weight"] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / fan_in)
        sd[f"{name}.bias"] = torch.randn(co, generator=g) * 0.01
    return sd


class MissingWeightsError(RuntimeError):
    """No checkpoint was given.  The reference downloads the official files (SPN:149, LGN:383); this container and the
    GPU boxes have no network, so a path is required — seeded synthetic weights are an explicit opt-in for tests and
    benchmarks (``allow_synthetic_weights``), never a silent default: features from random weights are garbage."""


def _no_weights(what: str, env: str, allow_synthetic: bool) -> None:
    if not allow_synthetic:
        raise MissingWeightsError(
            f"{what}: no checkpoint given - set '<section>.weights_path' (or the {env} environment variable) to the official file; "
            "pass 'allow_synthetic_weights: True' only for tests / synthetic benchmarks")


def load_superpoint_state_dict(path: str | None = None, seed: int = 1234, allow_synthetic: bool = False) -> Dict[str, torch.Tensor]:
    if path is None:
        _no_weights("SuperPoint (superpoint_v1.pth, SPN:149)", "DIM_SUPERPOINT_WEIGHTS", allow_synthetic)
        return synthetic_superpoint_state_dict(seed)
    sd = torch.load(str(Path(path)), map_location="cpu")
    missing = [f"{n}.{s}" for n, *_ in SP_LAYERS for s in ("weight", "bias") if f"{n}.{s}" not in sd]
    if missing:
        raise KeyError(f"SuperPoint checkpoint {path} lacks keys {missing}")
    return {k: v.float().contiguous() for k, v in sd.items()}


def lightglue_confidence_thresholds(n_layers: int = 9) -> torch.Tensor:
    """LGN:581-584: 0.8 + 0.1*exp(-4 i / L), clipped to [0, 1], stored as float32."""
    import numpy as np

    return torch.Tensor([float(np.clip(0.8 + 0.1 * np.exp(-4.0 * i / n_layers), 0, 1)) for i in range(n_layers)])


def synthetic_lightglue_state_dict(seed: int = 0, input_dim: int = 256, n_layers: int = 9, dim: int = 256,
                                   gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """nn.Linear-style uniform init in the official key layout (LGN:361-378).

    ``gain`` > 1 sharpens the synthetic network (larger logits) so that softmaxes are
    not uniform and the mutual-NN stage has something to decide in tests."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f, bias=True, prefix=""):
        bound = gain / math.sqrt(in_f)
        d = {prefix + "weight": (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound}
        if bias:
            d[prefix + "bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return d

    sd: Dict[str, torch.Tensor] = {}
    sd["confidence_thresholds"] = lightglue_confidence_thresholds(n_layers)
    if input_dim != dim:
        sd.update(lin(dim, input_dim, prefix="input_proj."))
    sd["posenc.Wr.weight"] = torch.randn(dim // 4 // 2, 2, generator=g)  # head_dim//2 x 2, std gamma^-2 = 1 (LGN:62-63)
    for i in range(n_layers):
        t = f"transformers.{i}."
        sd.update(lin(3 * dim, dim, prefix=t + "self_attn.Wqkv."))
        sd.update(lin(dim, dim, prefix=t + "self_attn.out_proj."))
        for blk in ("self_attn", "cross_attn"):
            sd.update(lin(2 * dim, 2 * dim, prefix=t + blk + ".ffn.0."))
            sd[t + blk + ".ffn.1.weight"] = 1.0 + 0.1 * torch.randn(2 * dim, generator=g)
            sd[t + blk + ".ffn.1.bias"] = 0.1 * torch.randn(2 * dim, generator=g)
            sd.update(lin(dim, 2 * dim, prefix=t + blk + ".ffn.3."))
        for nm in ("to_qk", "to_v", "to_out"):
            sd.update(lin(dim, dim, prefix=t + "cross_attn." + nm + "."))
        sd.update(lin(1, dim, prefix=f"log_assignment.{i}.matchability."))
        sd.update(lin(dim, dim, prefix=f"log_assignment.{i}.final_proj."))
        if i < n_layers - 1:
            sd.update(lin(1, dim, prefix=f"token_confidence.{i}.token.0."))
    return sd


def synthetic_lightglue_matching_state_dict(seed: int = 0, input_dim: int = 256, n_layers: int = 9, dim: int = 256,
                                            residual: float = 0.003, sharpness: float = 220.0, matchability_bias: float = 5.0,
                                            center: Optional[torch.Tensor] = None, whiten: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Seeded synthetic LightGlue weights (official key layout) that MATCH: the transformer blocks are near-identity (`ffn.3`
    scaled by ``residual``, so a keypoint's state stays close to its input descriptor), every `final_proj` is
    ``sharpness`` x (identity - ``center``) (an orthogonal map: the similarity is sharpness^2 / 16 x the inner product of the CENTRED
    descriptors, which a random `final_proj` does not preserve; ``center`` = the mean descriptor of the workload, passed as
    `final_proj.bias` = -sharpness x center — the seeded SuperPoint's descriptors have a pairwise cosine of 0.98, so without centring
    the logits would have to be ~6000 to separate them and the REFERENCE's own fp32 evaluation of the matching scores would carry
    1e-3 of noise) and the matchability heads say "matchable".  On two views that share keypoints with equal
    descriptors (workloads.shifted_crops) the mutual-NN assignment then returns the true correspondences — several hundred
    matches per pair with scores above the reference's default threshold 0.1 — so that verification, the match writers and the
    multi-GPU match gather of the benchmarks carry real work.  Same arithmetic, same FLOPs as any other weights.
    ``whiten`` ([dim, dim], symmetric; ``descriptor_whitening``): `final_proj` = sharpness x whiten x (x - center) instead of the scaled identity.
    DIFFERENT photographs share no equal descriptors, and in the plain centred dot product a few long descriptors are everyone's nearest
    neighbour (20 - 41 mutual nearest neighbours among 2000 x 2000 keypoints of the DSC photographs under the seeded SuperPoint); after whitening
    the similarity is a Mahalanobis one and 216 - 406 are mutual (sharpness 2: 132 - 180 matches above 0.1, logits <= 220)."""
    sd = synthetic_lightglue_state_dict(seed, input_dim, n_layers, dim, gain=1.0)
    for k in list(sd):
        if ".ffn.3." in k:
            sd[k] = sd[k] * residual
        elif k.endswith("final_proj.weight"):
            sd[k] = torch.eye(dim) * sharpness if whiten is None else (sharpness * whiten.detach().float().cpu()).contiguous()
        elif k.endswith("final_proj.bias"):
            c = None if center is None else center.detach().float().cpu().reshape(-1)
            sd[k] = torch.zeros_like(sd[k]) if c is None else (-sharpness * (c if whiten is None else whiten.detach().float().cpu() @ c)).contiguous()
        elif k.endswith("matchability.weight"):
            sd[k] = sd[k] * 0.1
        elif k.endswith("matchability.bias"):
            sd[k] = torch.full_like(sd[k], matchability_bias)
    return sd


def descriptor_whitening(desc_nd: torch.Tensor, eps: float = 1e-5):
    """(center [D], whiten [D, D] float32) of a descriptor sample [N, D]: the mean and the symmetric inverse square root of the covariance
    (+ eps), computed in float64 — the arguments of synthetic_lightglue_matching_state_dict(center=, whiten=)."""
    x = desc_nd.detach().double().cpu()
    c = x.mean(0)
    xc = x - c
    ev, u = torch.linalg.eigh(xc.t() @ xc / x.shape[0])
    return c.float(), (u @ torch.diag(1.0 / torch.sqrt(ev.clamp_min(0.0) + eps)) @ u.t()).float().contiguous()


def load_lightglue_state_dict(path: str | None = None, seed: int = 0, input_dim: int = 256, n_layers: int = 9,
                              gain: float = 1.0, allow_synthetic: bool = False) -> Dict[str, torch.Tensor]:
    if path is None:
        _no_weights("LightGlue (<features>_lightglue.pth, LGN:383)", "DIM_LIGHTGLUE_WEIGHTS", allow_synthetic)
        return synthetic_lightglue_state_dict(seed, input_dim, n_layers, gain=gain)
    sd = torch.load(str(Path(path)), map_location="cpu")
    for i in range(n_layers):  # legacy names (LGN:389-396)
        sd = {k.replace(f"self_attn.{i}", f"transformers.{i}.self_attn"): v for k, v in sd.items()}
        sd = {k.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn"): v for k, v in sd.items()}
    if "confidence_thresholds" not in sd:
        sd["confidence_thresholds"] = lightglue_confidence_thresholds(n_layers)
    return {k: v.float().contiguous() for k, v in sd.items()}


ALIKED_CFGS = {  # ALN:573-579  c1, c2, c3, c4, dim, K, M
    "aliked-t16": (8, 16, 32, 64, 64, 3, 16),
    "aliked-n16": (16, 32, 64, 128, 128, 3, 16),
    "aliked-n16rot": (16, 32, 64, 128, 128, 3, 16),
    "aliked-n32": (16, 32, 64, 128, 128, 3, 32),
}


def synthetic_aliked_state_dict(seed: int = 7, model_name: str = "aliked-n16rot") -> Dict[str, torch.Tensor]:
    """Seeded synthetic ALIKED weights in the official key layout (the real
    aliked-*.pth files live in the reference tree under thirdparty/ALIKED/models and load unchanged
    through load_aliked_state_dict).  BN running stats are included for layout fidelity only: the
    plugin runs BatchNorm in training mode (Q7) and never reads them."""
    c1, c2, c3, c4, dim, K, M = ALIKED_CFGS[model_name]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci, k, bias=False, gain=1.0):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (gain * math.sqrt(2.0 / (ci * k * k)))
        if bias:
            sd[name + ".bias"] = torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    conv("block1.conv1", c1, 3, 3); bn("block1.bn1", c1); conv("block1.conv2", c1, c1, 3); bn("block1.bn2", c1)
    conv("block2.conv1", c2, c1, 3); bn("block2.bn1", c2); conv("block2.conv2", c2, c2, 3); bn("block2.bn2", c2)
    conv("block2.downsample", c2, c1, 1, bias=True)
    for blk, ci, co in (("block3", c2, c3), ("block4", c3, c4)):
        conv(blk + ".conv1.offset_conv", 2 * K * K, ci, 3, bias=True, gain=0.7)
        conv(blk + ".conv1.regular_conv", co, ci, 3)
        bn(blk + ".bn1", co)
        conv(blk + ".conv2.offset_conv", 2 * K * K, co, 3, bias=True, gain=0.7)
        conv(blk + ".conv2.regular_conv", co, co, 3)
        bn(blk + ".bn2", co)
        conv(blk + ".downsample", co, ci, 1, bias=True)
    conv("conv1", dim // 4, c1, 1); conv("conv2", dim // 4, c2, 1); conv("conv3", dim // 4, c3, 1); conv("conv4", dim // 4, dim, 1)
    conv("score_head.0", 8, dim, 1); conv("score_head.2", 4, 8, 3); conv("score_head.4", 4, 4, 3); conv("score_head.6", 1, 4, 3, gain=2.0)
    sd["desc_head.agg_weights"] = torch.rand(M, dim, dim, generator=g) * (2.0 / math.sqrt(M * dim)) - (1.0 / math.sqrt(M * dim))
    conv("desc_head.offset_conv.0", 2 * M, dim, K, bias=True)
    conv("desc_head.offset_conv.2", 2 * M, 2 * M, 1, bias=True, gain=2.0)
    conv("desc_head.sf_conv", dim, dim, 1)
    return sd


def load_aliked_state_dict(path: str | None = None, seed: int = 7, model_name: str = "aliked-n16rot",
                           allow_synthetic: bool = False) -> Dict[str, torch.Tensor]:
    if path is None:
        _no_weights("ALIKED (thirdparty/ALIKED/models/aliked-*.pth in the reference tree)", "DIM_ALIKED_WEIGHTS", allow_synthetic)
        return synthetic_aliked_state_dict(seed, model_name)
    sd = torch.load(str(Path(path)), map_location="cpu")
    return {k: (v.float().contiguous() if v.is_floating_point() else v) for k, v in sd.items()}
