"""Thin host wrapper around the dim_lg_* C ABI (one resident matcher handle).

Host code allocates tensors and passes raw pointers; all compute is in libdim_hip.so.
``lib``/``device`` are injectable for the CPU emulator tests; the product default is the
gfx950 library on ``cuda`` and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import capi

_F = ctypes.c_void_p

_LAYER_FIELDS = [
    ("self_Wqkv_w", "transformers.{i}.self_attn.Wqkv.weight"), ("self_Wqkv_b", "transformers.{i}.self_attn.Wqkv.bias"),
    ("self_out_w", "transformers.{i}.self_attn.out_proj.weight"), ("self_out_b", "transformers.{i}.self_attn.out_proj.bias"),
    ("self_ffn0_w", "transformers.{i}.self_attn.ffn.0.weight"), ("self_ffn0_b", "transformers.{i}.self_attn.ffn.0.bias"),
    ("self_ln_w", "transformers.{i}.self_attn.ffn.1.weight"), ("self_ln_b", "transformers.{i}.self_attn.ffn.1.bias"),
    ("self_ffn3_w", "transformers.{i}.self_attn.ffn.3.weight"), ("self_ffn3_b", "transformers.{i}.self_attn.ffn.3.bias"),
    ("cross_qk_w", "transformers.{i}.cross_attn.to_qk.weight"), ("cross_qk_b", "transformers.{i}.cross_attn.to_qk.bias"),
    ("cross_v_w", "transformers.{i}.cross_attn.to_v.weight"), ("cross_v_b", "transformers.{i}.cross_attn.to_v.bias"),
    ("cross_out_w", "transformers.{i}.cross_attn.to_out.weight"), ("cross_out_b", "transformers.{i}.cross_attn.to_out.bias"),
    ("cross_ffn0_w", "transformers.{i}.cross_attn.ffn.0.weight"), ("cross_ffn0_b", "transformers.{i}.cross_attn.ffn.0.bias"),
    ("cross_ln_w", "transformers.{i}.cross_attn.ffn.1.weight"), ("cross_ln_b", "transformers.{i}.cross_attn.ffn.1.bias"),
    ("cross_ffn3_w", "transformers.{i}.cross_attn.ffn.3.weight"), ("cross_ffn3_b", "transformers.{i}.cross_attn.ffn.3.bias"),
    ("assign_match_w", "log_assignment.{i}.matchability.weight"), ("assign_match_b", "log_assignment.{i}.matchability.bias"),
    ("assign_proj_w", "log_assignment.{i}.final_proj.weight"), ("assign_proj_b", "log_assignment.{i}.final_proj.bias"),
    ("token_w", "token_confidence.{i}.token.0.weight"), ("token_b", "token_confidence.{i}.token.0.bias"),
]


class _LgLayer(ctypes.Structure):
    _fields_ = [(n, _F) for n, _ in _LAYER_FIELDS]


class _LgWeights(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int), ("input_dim", ctypes.c_int), ("input_proj_w", _F), ("input_proj_b", _F),
                ("posenc_Wr", _F), ("confidence_thresholds", _F), ("layers", ctypes.POINTER(_LgLayer))]


class _LgConfig(ctypes.Structure):
    _fields_ = [("depth_confidence", ctypes.c_double), ("width_confidence", ctypes.c_double),
                ("filter_threshold", ctypes.c_double), ("pruning_min_kpts", ctypes.c_int)]


class LightGlueHIP:
    """Resident LightGlue on one GPU.  conf keys follow LightGlue._default_conf (LGN:301-314)."""

    default_conf = {"n_layers": 9, "depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1,
                    "pruning_min_kpts": -1}

    def __init__(self, state_dict: Dict[str, torch.Tensor], conf: Optional[dict] = None, max_pairs: int = 1,
                 max_kpts: int = 2048, device="cuda", lib=None, on_saturation: str = "fallback", arithmetic=None):
        self.conf = {**self.default_conf, **(conf or {})}
        self.arithmetic = arithmetic        # None: the process default; "fp16x3" | "bf16x6" | "fp32": this handle only
        self.on_saturation = on_saturation  # fp16x3 range guard policy of __call__: "fallback" (bf16x6 re-run) | "raise" | "off"
        self.lib = lib if lib is not None else capi.load()
        self.device = torch.device(device)
        if lib is None and self.device.type != "cuda":
            raise capi.DimHipError("LightGlueHIP needs a HIP device; there is no CPU fallback")
        L = int(self.conf["n_layers"])
        keep = []

        def host(name):
            t = state_dict[name].detach().float().contiguous().cpu()
            keep.append(t)
            return t.data_ptr()

        self.input_dim = int(state_dict["input_proj.weight"].shape[1]) if "input_proj.weight" in state_dict else 256
        layers = (_LgLayer * L)()
        for i in range(L):
            for field, key in _LAYER_FIELDS:
                k = key.format(i=i)
                setattr(layers[i], field, host(k) if k in state_dict else None)
        w = _LgWeights()
        w.n_layers, w.input_dim = L, self.input_dim
        w.input_proj_w = host("input_proj.weight") if self.input_dim != 256 else None
        w.input_proj_b = host("input_proj.bias") if self.input_dim != 256 else None
        w.posenc_Wr = host("posenc.Wr.weight")
        w.confidence_thresholds = host("confidence_thresholds")
        w.layers = layers
        c = _LgConfig(float(self.conf["depth_confidence"]), float(self.conf["width_confidence"]),
                      float(self.conf["filter_threshold"]), int(self.conf["pruning_min_kpts"]))
        self.max_pairs = int(max_pairs)
        self._h = ctypes.c_void_p()
        with self._ctx():
            capi.check(self.lib, self.lib.dim_lg_create(ctypes.byref(w), ctypes.byref(c), self.max_pairs, int(max_kpts), ctypes.byref(self._h)))
        self.nk = self.lib.dim_lg_max_kpts(self._h)
        if arithmetic is not None:
            capi.set_handle_arithmetic(self.lib, self._h, arithmetic)
        del keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.dim_lg_destroy(h)
            self._h = None

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _ctx(self):
        """The library launches on the CURRENT HIP device: make it the handle's."""
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    def match_batch_guarded(self, *a, logger=None, **k):
        """match_batch under the fp16x3 range guard (capi.run_guarded): synchronises."""
        with self._ctx():
            return capi.run_guarded(self.lib, self._stream(), lambda: self.match_batch(*a, **k), "LightGlue", self.on_saturation, logger, handle=self._h, arithmetic=self.arithmetic)

    @torch.no_grad()
    def match_batch(self, kpts_tab, desc_tab, n_tab, size_tab, pair_idx=None, n_pairs=None, dense=False, out=None):
        """Device feature table -> device match tables; no host sync.  See dim_hip.h:dim_lg_match."""
        cap = kpts_tab.shape[1]
        assert kpts_tab.dtype == torch.float32 and desc_tab.dtype == torch.float32 and n_tab.dtype == torch.int32
        assert kpts_tab.is_contiguous() and desc_tab.is_contiguous() and size_tab.is_contiguous()
        assert desc_tab.shape[1] == cap and desc_tab.shape[2] == self.input_dim
        if pair_idx is not None:
            assert pair_idx.dtype == torch.int32 and pair_idx.is_contiguous()
            P = pair_idx.shape[0] if n_pairs is None else n_pairs
        else:
            P = kpts_tab.shape[0] // 2 if n_pairs is None else n_pairs
        dev, NK = kpts_tab.device, self.nk
        if out is None:
            out = {
                "matches": torch.empty(P, NK, 2, dtype=torch.int64, device=dev),
                "scores": torch.empty(P, NK, dtype=torch.float32, device=dev),
                "n_matches": torch.zeros(P, dtype=torch.int32, device=dev),
                "matches01": torch.empty(P, 2, NK, dtype=torch.int32, device=dev),
                "mscores01": torch.empty(P, 2, NK, dtype=torch.float32, device=dev),
                "stop": torch.zeros(P, dtype=torch.int32, device=dev),
                "prune01": torch.empty(P, 2, NK, dtype=torch.int32, device=dev),
            }
        if dense:
            out["dense"] = torch.zeros(P, NK + 1, NK + 1, dtype=torch.float32, device=dev)
        with self._ctx():
            capi.check(self.lib, self.lib.dim_lg_match(
                self._h, capi.ptr(kpts_tab), capi.ptr(desc_tab), capi.ptr(n_tab), capi.ptr(size_tab), int(cap),
                capi.ptr(pair_idx), int(P), capi.ptr(out["matches"]), capi.ptr(out["scores"]), capi.ptr(out["n_matches"]),
                capi.ptr(out["matches01"]), capi.ptr(out["mscores01"]), capi.ptr(out["stop"]), capi.ptr(out["prune01"]),
                capi.ptr(out.get("dense")), self._stream()))
        return out

    @torch.no_grad()
    def __call__(self, data: dict, dense: bool = False) -> dict:
        """Reference-style call for ONE pair (LGN:415-579): data = {"image0": {keypoints [1,M,2],
        descriptors [1,M,D], image_size [1,2]}, "image1": {...}}.  Returns the reference's dict with
        the batch dimension kept at 1 for matches0/1 etc. and lists for matches/scores."""
        d0, d1 = data["image0"], data["image1"]
        k0, k1 = d0["keypoints"][0], d1["keypoints"][0]
        m, n = k0.shape[0], k1.shape[0]
        cap = max(m, n, 1)
        dev = self.device
        kt = torch.zeros(2, cap, 2, dtype=torch.float32, device=dev)
        dt = torch.zeros(2, cap, self.input_dim, dtype=torch.float32, device=dev)
        kt[0, :m], kt[1, :n] = k0.to(dev, torch.float32), k1.to(dev, torch.float32)
        dt[0, :m], dt[1, :n] = d0["descriptors"][0].to(dev, torch.float32), d1["descriptors"][0].to(dev, torch.float32)
        nt = torch.tensor([m, n], dtype=torch.int32, device=dev)
        st = torch.stack([d0["image_size"][0].float(), d1["image_size"][0].float()]).to(dev).contiguous()
        o = self.match_batch_guarded(kt, dt, nt, st, n_pairs=1, dense=dense)
        S = int(o["n_matches"][0].item())
        res = {
            "matches0": o["matches01"][0, 0, :m].long()[None], "matches1": o["matches01"][0, 1, :n].long()[None],
            "matching_scores0": o["mscores01"][0, 0, :m][None], "matching_scores1": o["mscores01"][0, 1, :n][None],
            "stop": int(o["stop"][0].item()), "matches": [o["matches"][0, :S]], "scores": [o["scores"][0, :S]],
            "prune0": o["prune01"][0, 0, :m][None], "prune1": o["prune01"][0, 1, :n][None],
        }
        if dense:
            res["dense"] = o["dense"][0]
        return res
