"""Adversarial inputs for the fp16x3 arithmetic: weight / activation ranges that seeded He-normal
weights and uniform-noise images never reach.  Shared by the emulator (CPU) and the GPU tests.

Every case keeps the NETWORK FUNCTION comparable to the oracle (the oracle gets the same modified weights); what changes
is the range of the intermediate tensors the split-precision kernels see:

  rescale_xN   conv2a weights+bias x N, conv2b weights / N — the function is unchanged, the activation between them is N
               times larger (N = 64 stays inside the exact range; N = 4096 leaves it: the range guard must fire and the
               plugin-level call must come back correct through the bf16x6 re-run)
  outlier      one conv3a weight multiplied by 100 (per-tensor scaling would push every other weight towards the
               subnormal low piece; scales are per output channel)
  channel_mix  output channels of conv1b with magnitudes 2^-10 .. 2^10 (compensated in conv2a's input channels)
  zeros        image == 0 (every activation is a bias chain)
  tiny         image * 1e-4 (all first-layer activations far below the point where the low fp16 piece turns subnormal)
"""
from __future__ import annotations

import importlib

import torch

weights = importlib.import_module("deep-image-matching_amd.weights")


def sp_case(name: str, H: int, W: int, seed: int = 3):
    """-> (state_dict, image [1,1,H,W], expect_guard: bool)"""
    sd = {k: v.clone() for k, v in weights.synthetic_superpoint_state_dict(1234).items()}
    img = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(seed))
    guard = False
    if name.startswith("rescale_x"):
        n = float(name[len("rescale_x"):])
        sd["conv2a.weight"] *= n
        sd["conv2a.bias"] *= n
        sd["conv2b.weight"] /= n
        guard = n >= 2048
    elif name == "outlier":
        sd["conv3a.weight"][5, 7, 1, 1] *= 100.0
    elif name == "channel_mix":
        s = torch.pow(2.0, torch.linspace(-10, 10, 64).round())
        sd["conv1b.weight"] *= s[:, None, None, None]
        sd["conv1b.bias"] *= s
        sd["conv2a.weight"] /= s[None, :, None, None]
    elif name == "zeros":
        img = torch.zeros_like(img)
    elif name == "tiny":
        img = img * 1e-4
    elif name == "bright":
        img = img * 3.0  # an image that was not scaled to [0, 1]: |image| > 1 trips the input guard
        guard = True
    else:
        raise KeyError(name)
    return sd, img, guard


SP_CASES = ["rescale_x64", "rescale_x4096", "outlier", "channel_mix", "zeros", "tiny", "bright"]


def lg_case(name: str, m: int = 96, n: int = 80, seed: int = 5, n_layers: int = 3):
    """-> (state_dict, (kpts0, desc0, size0), (kpts1, desc1, size1), conf, expect_guard)"""
    from tests import golden_cases as gc

    case = dict(gc.LG_CASES["fixed"], m=m, n=n, seed=seed)
    f0, f1 = gc.lg_inputs(case)
    sd = weights.synthetic_lightglue_state_dict(1, 256, n_layers=n_layers, gain=2.0)
    conf = {"n_layers": n_layers, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0, "pruning_min_kpts": -1}
    guard = False
    if name == "desc_1e5":  # un-normalised descriptors: the input itself leaves the range
        f0["desc"], f1["desc"] = f0["desc"] * 1e5, f1["desc"] * 1e5
        guard = True
    elif name == "ffn_x512":  # ffn.0 x 512 (LayerNorm removes the scale again): a huge pre-LN tensor, fp32 only, must NOT trip
        for i in range(n_layers):
            for blk in ("self_attn", "cross_attn"):
                sd[f"transformers.{i}.{blk}.ffn.0.weight"] *= 512.0
                sd[f"transformers.{i}.{blk}.ffn.0.bias"] *= 512.0
    elif name == "v_x4096":  # v x 4096 and out_proj / 4096: the function is unchanged, the value projection leaves the range
        for i in range(n_layers):
            rows = torch.arange(768) % 3 == 2   # Wqkv rows are head*192 + dim*3 + {q,k,v} (LGN:153-154)
            sd[f"transformers.{i}.self_attn.Wqkv.weight"][rows] *= 4096.0
            sd[f"transformers.{i}.self_attn.Wqkv.bias"][rows] *= 4096.0
            sd[f"transformers.{i}.self_attn.out_proj.weight"] /= 4096.0
        guard = True
    elif name == "hidden_x8192":  # LayerNorm's affine x 8192 and ffn.3 / 8192: GELU is not homogeneous, so the function changes, but the point is
        # the RANGE: the hidden tensor (what ffn.3's split consumes) leaves +-4094 and the guard of whichever kernel produces it must fire
        for i in range(n_layers):
            for blk in ("self_attn", "cross_attn"):
                sd[f"transformers.{i}.{blk}.ffn.1.weight"] *= 8192.0
                sd[f"transformers.{i}.{blk}.ffn.1.bias"] *= 8192.0
                sd[f"transformers.{i}.{blk}.ffn.3.weight"] /= 8192.0
        guard = True
    elif name == "outlier":
        sd["transformers.0.self_attn.ffn.3.weight"][3, 11] *= 100.0
    elif name == "tiny_desc":
        f0["desc"], f1["desc"] = f0["desc"] * 1e-2, f1["desc"] * 1e-2
    else:
        raise KeyError(name)
    return sd, f0, f1, conf, guard


LG_CASES = ["desc_1e5", "ffn_x512", "v_x4096", "hidden_x8192", "outlier", "tiny_desc"]
