"""Thin host wrapper around the dim_sp_* C ABI (one resident extractor handle).

Host code only allocates tensors and passes raw pointers; all compute is in
libdim_hip.so.  ``lib``/``device`` are injectable so the CPU tests can drive the very
same sources through the test-only emulator build; the product default is the
gfx950 library on ``cuda`` and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import capi
from .weights import SP_LAYERS


class _SpWeights(ctypes.Structure):
    _fields_ = [("conv_w", ctypes.c_void_p * 12), ("conv_b", ctypes.c_void_p * 12)]


class _SpConfig(ctypes.Structure):
    _fields_ = [
        ("nms_radius", ctypes.c_int),
        ("keypoint_threshold", ctypes.c_float),
        ("max_keypoints", ctypes.c_int),
        ("remove_borders", ctypes.c_int),
        ("fix_sampling", ctypes.c_int),
    ]


class SuperPointHIP:
    """Resident SuperPoint on one GPU.  cfg keys follow SPN:112-118 (+ fix_sampling)."""

    default_config = {"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": -1, "remove_borders": 4,
                      "fix_sampling": False}

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: Optional[dict] = None, max_batch: int = 1,
                 max_hw=(1024, 1024), capacity: Optional[int] = None, device="cuda", lib=None, on_saturation: str = "fallback", arithmetic=None):
        self.cfg = {**self.default_config, **(cfg or {})}
        self.arithmetic = arithmetic        # None: the process default (capi.set_arithmetic); "fp16x3" | "bf16x6" | "fp32": this handle only
        self.on_saturation = on_saturation  # fp16x3 range guard policy of __call__: "fallback" (bf16x6 re-run) | "raise" | "off"
        mk = self.cfg["max_keypoints"]
        if mk == 0 or mk < -1:
            raise ValueError('"max_keypoints" must be positive or "-1"')  # SPN:152-154
        self.lib = lib if lib is not None else capi.load()
        self.device = torch.device(device)
        if lib is None and self.device.type != "cuda":
            raise capi.DimHipError("SuperPointHIP needs a HIP device; there is no CPU fallback")
        self.max_batch, self.max_hw = int(max_batch), (int(max_hw[0]), int(max_hw[1]))
        self.capacity = int(capacity if capacity is not None else (mk if mk > 0 else 8192))
        w = _SpWeights()
        keep = []
        for i, (name, *_ ) in enumerate(SP_LAYERS):
            wt = state_dict[name + ".weight"].detach().float().contiguous().cpu()
            bt = state_dict[name + ".bias"].detach().float().contiguous().cpu()
            keep += [wt, bt]
            w.conv_w[i] = wt.data_ptr()
            w.conv_b[i] = bt.data_ptr()
        c = _SpConfig(int(self.cfg["nms_radius"]), float(self.cfg["keypoint_threshold"]), int(mk),
                      int(self.cfg["remove_borders"]), int(bool(self.cfg["fix_sampling"])))
        self._h = ctypes.c_void_p()
        with self._ctx():
            capi.check(self.lib, self.lib.dim_sp_create(ctypes.byref(w), ctypes.byref(c), self.max_batch, self.max_hw[0],
                                                        self.max_hw[1], self.capacity, ctypes.byref(self._h)))
        if arithmetic is not None:
            capi.set_handle_arithmetic(self.lib, self._h, arithmetic)
        del keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.dim_sp_destroy(h)
            self._h = None

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _ctx(self):
        """The library launches on the CURRENT HIP device: make it the handle's."""
        return torch.cuda.device(self.device) if self.device.type == "cuda" else _null()

    def candidate_counts(self, batch: int) -> torch.Tensor:
        """NMS survivors above threshold / border per image of the last call, BEFORE top-k / the capacity
        cut (host int32 tensor; synchronises).  > capacity in keep-all mode (max_keypoints = -1) means the call
        dropped keypoints the reference would return: re-create the handle with a larger capacity."""
        p = ctypes.c_void_p()
        capi.check(self.lib, self.lib.dim_sp_candidate_counts(self._h, ctypes.byref(p)))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            out = torch.empty(batch, dtype=torch.int32, device=self.device)
            rc = ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(out.data_ptr()), p, ctypes.c_size_t(batch * 4), 3)
            if rc != 0:
                raise capi.DimHipError(f"hipMemcpy failed: {rc}")
            return out.cpu()
        return torch.frombuffer((ctypes.c_int32 * batch).from_address(p.value), dtype=torch.int32).clone()

    @torch.no_grad()
    def extract_batch(self, images: torch.Tensor, out=None):
        """images [B,H,W] float32 in [0,1] on self.device.  Returns device tensors
        (kpts [B,cap,2], scores [B,cap], desc [B,cap,256], n [B] int32); no host sync.
        ``out`` = a previously returned tuple to write into (no allocation: needed when the call is
        issued on a side stream)."""
        assert images.dim() == 3 and images.dtype == torch.float32 and images.is_contiguous()
        B, H, W = images.shape
        dev = images.device
        if out is not None:
            kp, sc, de, n = out
        else:
            kp = torch.empty(B, self.capacity, 2, dtype=torch.float32, device=dev)
            sc = torch.empty(B, self.capacity, dtype=torch.float32, device=dev)
            de = torch.empty(B, self.capacity, 256, dtype=torch.float32, device=dev)
            n = torch.zeros(B, dtype=torch.int32, device=dev)
        with self._ctx():
            capi.check(self.lib, self.lib.dim_sp_extract(self._h, capi.ptr(images), B, H, W, capi.ptr(kp), capi.ptr(sc),
                                                         capi.ptr(de), capi.ptr(n), self._stream()))
        return kp, sc, de, n

    def extract_batch_guarded(self, images: torch.Tensor, out=None, logger=None):
        """extract_batch under the fp16x3 range guard (capi.run_guarded): synchronises."""
        with self._ctx():
            return capi.run_guarded(self.lib, self._stream(), lambda: self.extract_batch(images, out=out), "SuperPoint",
                                    self.on_saturation, logger, handle=self._h, arithmetic=self.arithmetic)

    @torch.no_grad()
    def __call__(self, image: torch.Tensor) -> dict:
        """image [1,1,H,W] (the reference's input, SPN:158).  Returns the reference's dict
        for one image with tensors on the device: keypoints (N,2), scores (N,), descriptors (256,N)."""
        img = image.reshape(image.shape[-2], image.shape[-1])[None].contiguous().to(self.device, torch.float32)
        kp, sc, de, n = self.extract_batch_guarded(img)
        k = int(n[0].item())
        return {"keypoints": kp[0, :k], "scores": sc[0, :k], "descriptors": de[0, :k].t()}

    def debug_taps(self, batch: int = 1) -> dict:
        """Intermediate tensors of the last call as CPU copies (parity tests)."""
        ptrs = [ctypes.c_void_p() for _ in range(5)]
        h8, w8 = ctypes.c_int(), ctypes.c_int()
        capi.check(self.lib, self.lib.dim_sp_debug_buffers(self._h, *[ctypes.byref(p) for p in ptrs], ctypes.byref(h8), ctypes.byref(w8)))
        H8, W8 = h8.value, w8.value
        h, w = H8 // 8, W8 // 8
        shapes = {"encoder": (batch, h, w, 128), "logits": (batch, h * w, 65), "score_map": (batch, H8, W8),
                  "nms_map": (batch, H8, W8), "dense_desc": (batch, h, w, 256)}
        out = {}
        for (name, shape), p in zip(shapes.items(), ptrs):
            out[name] = _copy_from(self.lib, p.value, shape, self.device)
        return out


    def debug_conv1b(self, batch: int, H: int, W: int) -> torch.Tensor:
        """conv1b's pooled output of the last call as a CPU tensor [batch, H/2, W/2, 64] (fp32 NHWC): dim_sp_debug_conv1b."""
        p, h2, w2 = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        capi.check(self.lib, self.lib.dim_sp_debug_conv1b(self._h, int(batch), int(H), int(W), ctypes.byref(p), ctypes.byref(h2), ctypes.byref(w2)))
        return _copy_from(self.lib, p.value, (batch, h2.value, w2.value, 64), self.device)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _copy_from(lib, addr: int, shape, device) -> torch.Tensor:
    """Copy a raw device buffer into a CPU tensor (hipMemcpy through torch)."""
    import math

    n = math.prod(shape)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
        out = torch.empty(n, dtype=torch.float32, device=device)
        rc = ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(n * 4), 3)
        if rc != 0:
            raise capi.DimHipError(f"hipMemcpy failed: {rc}")
        return out.cpu().reshape(shape)
    buf = (ctypes.c_float * n).from_address(addr)
    return torch.frombuffer(buf, dtype=torch.float32).clone().reshape(shape)
