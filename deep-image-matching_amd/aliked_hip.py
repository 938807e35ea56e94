"""Thin host wrapper around the dim_aliked_* C ABI (one resident ALIKED extractor handle)."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import capi
from .weights import ALIKED_CFGS

_F = ctypes.c_void_p
_W_FIELDS = [
    ("block1_conv1", "block1.conv1.weight"), ("block1_conv2", "block1.conv2.weight"),
    ("block2_conv1", "block2.conv1.weight"), ("block2_conv2", "block2.conv2.weight"),
    ("block2_ds_w", "block2.downsample.weight"), ("block2_ds_b", "block2.downsample.bias"),
    ("block3_off1_w", "block3.conv1.offset_conv.weight"), ("block3_off1_b", "block3.conv1.offset_conv.bias"),
    ("block3_reg1", "block3.conv1.regular_conv.weight"),
    ("block3_off2_w", "block3.conv2.offset_conv.weight"), ("block3_off2_b", "block3.conv2.offset_conv.bias"),
    ("block3_reg2", "block3.conv2.regular_conv.weight"),
    ("block3_ds_w", "block3.downsample.weight"), ("block3_ds_b", "block3.downsample.bias"),
    ("block4_off1_w", "block4.conv1.offset_conv.weight"), ("block4_off1_b", "block4.conv1.offset_conv.bias"),
    ("block4_reg1", "block4.conv1.regular_conv.weight"),
    ("block4_off2_w", "block4.conv2.offset_conv.weight"), ("block4_off2_b", "block4.conv2.offset_conv.bias"),
    ("block4_reg2", "block4.conv2.regular_conv.weight"),
    ("block4_ds_w", "block4.downsample.weight"), ("block4_ds_b", "block4.downsample.bias"),
]
_BN = ["block1.bn1", "block1.bn2", "block2.bn1", "block2.bn2", "block3.bn1", "block3.bn2", "block4.bn1", "block4.bn2"]
_TAIL = [
    ("conv1", "conv1.weight"), ("conv2", "conv2.weight"), ("conv3", "conv3.weight"), ("conv4", "conv4.weight"),
    ("score0", "score_head.0.weight"), ("score2", "score_head.2.weight"), ("score4", "score_head.4.weight"), ("score6", "score_head.6.weight"),
    ("desc_off0_w", "desc_head.offset_conv.0.weight"), ("desc_off0_b", "desc_head.offset_conv.0.bias"),
    ("desc_off2_w", "desc_head.offset_conv.2.weight"), ("desc_off2_b", "desc_head.offset_conv.2.bias"),
    ("desc_sf", "desc_head.sf_conv.weight"), ("desc_agg", "desc_head.agg_weights"),
]


class _AlWeights(ctypes.Structure):
    _fields_ = [(n, _F) for n, _ in _W_FIELDS] + [("bn_weight", _F * 8), ("bn_bias", _F * 8)] + [(n, _F) for n, _ in _TAIL]


N_LIMIT_MAX = 20000   # ALN:571: DKD's n_limit when max_num_keypoints <= 0 (the keep-all modes)


class _AlConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("c1", "c2", "c3", "c4", "dim", "K", "M", "max_num_keypoints")] + \
               [("detection_threshold", ctypes.c_double), ("nms_radius", ctypes.c_int)]


class AlikedHIP:
    """Resident ALIKED on one GPU.  cfg keys follow ALIKED._default_conf (ALN:562-567)."""

    default_config = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 2}

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[dict] = None, max_batch: int = 1, max_hw=(1024, 1024),
                 capacity: Optional[int] = None, device="cuda", lib=None):
        self.cfg = {**self.default_config, **(cfg or {})}
        self.on_saturation = self.cfg.pop("on_saturation", "fallback")
        self.arithmetic = self.cfg.pop("arithmetic", None)   # None: the process default; "fp16x3" | "fp32": this handle only
        self.lib = lib if lib is not None else capi.load()
        self.device = torch.device(device)
        if lib is None and self.device.type != "cuda":
            raise capi.DimHipError("AlikedHIP needs a HIP device; there is no CPU fallback")
        keep = []

        def host(name):
            t = state_dict[name].detach().float().contiguous().cpu()
            keep.append(t)
            return t.data_ptr()

        w = _AlWeights()
        for f, k in _W_FIELDS + _TAIL:
            setattr(w, f, host(k))
        for i, b in enumerate(_BN):
            w.bn_weight[i] = host(b + ".weight")
            w.bn_bias[i] = host(b + ".bias")
        geo = ALIKED_CFGS[self.cfg["model_name"]]
        self.dim = int(geo[4])     # descriptor length: 128, or 64 for aliked-t16
        mk = int(self.cfg["max_num_keypoints"])
        self.capacity = int(capacity if capacity is not None else (mk if mk > 0 else N_LIMIT_MAX))
        c = _AlConfig(*geo, mk, float(self.cfg["detection_threshold"]), int(self.cfg["nms_radius"]))
        self.max_batch, self.max_hw = int(max_batch), (int(max_hw[0]), int(max_hw[1]))
        self._h = ctypes.c_void_p()
        with self._ctx():
            capi.check(self.lib, self.lib.dim_aliked_create(ctypes.byref(w), ctypes.byref(c), self.max_batch, self.max_hw[0], self.max_hw[1],
                                                            self.capacity, ctypes.byref(self._h)))
        if self.arithmetic is not None:
            capi.set_handle_arithmetic(self.lib, self._h, self.arithmetic)
        del keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.dim_aliked_destroy(h)
            self._h = None

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _ctx(self):
        """The library launches on the CURRENT HIP device: make it the handle's."""
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    @torch.no_grad()
    def extract_batch(self, images: torch.Tensor, out=None):
        """images [B,H,W,C] float32 in [0,1] (HWC, C = 3 or 1) on self.device -> device tensors
        (kpts [B,cap,2], scores [B,cap], desc [B,cap,dim], n [B] int32); no host sync.  ``out`` = a tuple of such tensors to write into
        (contiguous views of a larger buffer: pipeline.PairMatchingPipeline's exchange buffer)."""
        assert images.dim() == 4 and images.dtype == torch.float32 and images.is_contiguous()
        B, H, W, C = images.shape
        dev = images.device
        if out is not None:
            kp, sc, de, n = out
            with self._ctx():
                capi.check(self.lib, self.lib.dim_aliked_extract(self._h, capi.ptr(images), B, H, W, C, capi.ptr(kp), capi.ptr(sc), capi.ptr(de),
                                                                 capi.ptr(n), self._stream()))
            return kp, sc, de, n
        kp = torch.empty(B, self.capacity, 2, dtype=torch.float32, device=dev)
        sc = torch.empty(B, self.capacity, dtype=torch.float32, device=dev)
        de = torch.empty(B, self.capacity, self.dim, dtype=torch.float32, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        with self._ctx():
            capi.check(self.lib, self.lib.dim_aliked_extract(self._h, capi.ptr(images), B, H, W, C, capi.ptr(kp), capi.ptr(sc), capi.ptr(de),
                                                             capi.ptr(n), self._stream()))
        return kp, sc, de, n

    def extract_batch_guarded(self, images: torch.Tensor, logger=None):
        """extract_batch under the fp16x3 range guard (capi.run_guarded): the full- and half-resolution convolutions and the GEMMs
        run as fp16 splits on the matrix cores (aliked_x3.hip, gemm_x6.hip), exact for |activation| <= 4094; a call that
        leaves that range is repeated on the fp32 paths.  Synchronises."""
        with self._ctx():
            return capi.run_guarded(self.lib, self._stream(), lambda: self.extract_batch(images), "ALIKED", self.on_saturation, logger, handle=self._h, arithmetic=self.arithmetic)

    @torch.no_grad()
    def __call__(self, image: torch.Tensor) -> dict:
        """image [1,C,H,W] (the reference's input).  Returns DIM's feature dict for one image (device
        tensors): keypoints (N,2), descriptors (dim,N), scores (N,) (= dispersities, Q8)."""
        img = image[0].permute(1, 2, 0).contiguous().to(self.device, torch.float32)[None]
        kp, sc, de, n = self.extract_batch_guarded(img)
        k = int(n[0].item())
        return {"keypoints": kp[0, :k], "scores": sc[0, :k], "descriptors": de[0, :k].t()}

    def debug_taps(self, batch: int = 1) -> dict:
        from .superpoint_hip import _copy_from

        p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
        v = [ctypes.c_int() for _ in range(4)]
        capi.check(self.lib, self.lib.dim_aliked_debug_buffers(self._h, ctypes.byref(p1), ctypes.byref(p2), *[ctypes.byref(x) for x in v]))
        hp, wp, pt, pl = [x.value for x in v]
        return {"x1234": _copy_from(self.lib, p1.value, (batch, hp, wp, self.dim), self.device), "pad": (pt, pl), "hp_wp": (hp, wp),
                "score_ptr": p2.value}
