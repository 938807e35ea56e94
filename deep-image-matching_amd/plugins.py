"""Drop-in plugins for deep-image-matching's extractor / matcher plugin surface.

``SuperPointExtractor`` mirrors extractors/superpoint.py:64-146 and ``LightGlueMatcher``
mirrors matchers/lightglue.py:77-125: same class attributes (``_default_conf``,
``required_inputs``, ``grayscale``, ``descriptor_size`` …), same constructor arguments, same
``_extract(image) -> dict`` / ``_match_pairs(feats0, feats1) -> np.ndarray`` contracts and
error behaviour, with the network replaced by the gfx950 library.  When the real
``deep_image_matching`` package is importable the classes subclass its ``ExtractorBase`` /
``MatcherBase`` (so ``extractor_loader`` / ``matcher_loader`` discover them, extractor_base.py:29-52,
matcher_base.py:36-60); otherwise a minimal stand-in base with the same constructor contract
(extractor_base.py:119-160) is used so the hooks can be driven and tested on their own.
See INTEGRATION.md for the module files a maintainer adds to the reference tree.
"""
from __future__ import annotations

import logging
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from . import capi
from . import weights as _weights
from .aliked_hip import AlikedHIP
from .lightglue_hip import LightGlueHIP
from .superpoint_hip import SuperPointHIP
from .tile_matching import BatchedTileMatchingMixin
from .tiling import BatchedTilingMixin

logger = logging.getLogger("dim")

try:  # pragma: no cover - the reference package is not importable in the build container
    from deep_image_matching.extractors.extractor_base import ExtractorBase as _ExtractorBase
    from deep_image_matching.matchers.matcher_base import MatcherBase as _MatcherBase

    HAVE_DIM = True
except Exception:  # noqa: BLE001
    HAVE_DIM = False

    class _StandInBase:
        """Constructor contract of ExtractorBase/MatcherBase (extractor_base.py:119-160,
        matcher_base.py:95-140): config exposes .general/.extractor/.matcher (or is a dict with
        those keys); self.config = {"general", "extractor"|"matcher"}; device honours force_cpu."""

        _role = "extractor"
        _default_conf: dict = {}

        def __init__(self, config):
            get = (lambda k: config.get(k, {})) if isinstance(config, dict) else (lambda k: getattr(config, k, {}) or {})
            if not isinstance(config, dict) and not all(hasattr(config, a) for a in ("general",)):
                raise TypeError("`config` must be a Config object (or a dict with 'general' / '%s')" % self._role)
            general = dict(get("general"))
            self.config = {"general": general, self._role: {**self._default_conf, **dict(get(self._role))}}
            self._device = "cuda" if torch.cuda.is_available() and not general.get("force_cpu", False) else "cpu"

    class _ExtractorBase(_StandInBase):
        _role = "extractor"

    class _MatcherBase(_StandInBase):
        _role = "matcher"


def _apply_arithmetic(conf: dict, lib):
    """Optional plugin option ``arithmetic``: "fp16x3" (library default: 2-way fp16 splits x 3 MFMA terms; activations
    exact up to |x| = 4094 — guarded: a call that leaves the range is repeated in bf16x6, option ``on_saturation``),
    "bf16x6" (exact 3-way bf16 splits x 6 terms, no range limit) or "fp32" (plain fp32 MFMA).
    The choice belongs to THIS plugin's library handles (dim_handle_tune_set key 1, set when a handle is created); without the option
    the handles follow the process default (capi.set_arithmetic).  Returns the validated name or None."""
    name = (conf or {}).get("arithmetic")
    if name is None:
        return None
    if name not in capi.ARITHMETIC:
        raise ValueError(f"arithmetic must be one of {sorted(capi.ARITHMETIC)}, got {name!r}")
    return name


def _saturation_policy(conf: dict) -> str:
    pol = (conf or {}).get("on_saturation", "fallback")
    if pol not in ("fallback", "raise", "off"):
        raise ValueError(f"on_saturation must be 'fallback', 'raise' or 'off', got {pol!r}")
    return pol


class _PinnedStaging:
    """Page-locked host staging for the per-call hooks (one image in, one feature set out per call).  The reference's
    ``torch.tensor(image / 255.0).to(device)`` / ``.cpu().numpy()`` go through pageable memory; on the MI355X box that cost 5 - 15 ms per
    call around 0.8 ms of kernels (profiles/r05_hook_profile.txt).  Buffers grow on demand and are reused: a call must have finished with
    them (the hooks synchronise before they return) before the next call touches them."""

    def __init__(self):
        self._buf = {}

    def get(self, tag: str, nbytes: int) -> torch.Tensor:
        b = self._buf.get(tag)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
            self._buf[tag] = b
        return b

    def upload_scaled(self, image: np.ndarray, device) -> torch.Tensor:
        """``image / 255.0`` as float32, computed by numpy STRAIGHT INTO the page-locked buffer (one pass, no torch CPU op: a 256-thread intra-op
        pool needs milliseconds to wake up for a 4 MB copy), then an asynchronous copy to the device on the current stream.  Same values as the
        reference's torch.tensor(image / 255.0, dtype=torch.float): numpy divides in the input's precision and the result is rounded to float32."""
        n = int(image.size) * 4
        pin = self.get("in", n)[:n].view(torch.float32).view(tuple(image.shape))
        np.divide(image, 255.0, out=pin.numpy(), casting="same_kind")
        return pin.to(device, non_blocking=True)

    def download(self, tensors, device) -> list:
        """device tensors -> owned numpy arrays through ONE pinned buffer and ONE synchronisation."""
        sizes = [t.numel() * t.element_size() for t in tensors]
        off = [0]
        for z in sizes:
            off.append(off[-1] + ((z + 255) & ~255))
        pin = self.get("out", off[-1])
        views = []
        for t, o, z in zip(tensors, off, sizes):
            v = pin[o:o + z].view(t.dtype).view(t.shape)
            v.copy_(t, non_blocking=True)
            views.append(v)
        torch.cuda.current_stream(device).synchronize()
        return [v.numpy().copy() for v in views]


def _frame2array(image: np.ndarray) -> np.ndarray:
    """The layout half of the reference's _frame2tensor (SPX:134-146, ALX:66-78): (H, W) -> [1, 1, H, W], (H, W, C) -> [1, C, H, W] (a view)."""
    if len(image.shape) == 2:
        return image[None][None]
    if len(image.shape) == 3:
        return image.transpose(2, 0, 1)[None]
    return image


def _resolve_device(device, what: str):
    """The plugin's device: the reference's ``self._device`` ("cuda" unless general.force_cpu).  There is no CPU
    path; the CPU tests install the emulator build through capi.install(), which also names the device."""
    forced = capi.installed_device()
    if forced is not None:
        return forced
    kind = getattr(device, "type", str(device))
    if not str(kind).startswith("cuda"):
        raise RuntimeError(f"{what}: the MI355X plugin has no CPU path (device={device!r}); "
                           "use the reference's own plugin when general.force_cpu is set")
    return device


class SuperPointExtractor(BatchedTilingMixin, _ExtractorBase):
    """extractors/superpoint.py:64 — SuperPoint on the gfx950 library."""

    _default_conf = {
        "name": "superpoint",
        "nms_radius": 4,
        "keypoint_threshold": 0.005,
        "max_keypoints": -1,
        "remove_borders": 4,
        "fix_sampling": False,
    }
    required_inputs = ["image"]
    grayscale = True
    as_float = True
    descriptor_size = 256
    features_as_half = True
    detection_noise = 2.0

    def __init__(self, config):
        super().__init__(config)
        self._lib = capi.load()
        self._device = _resolve_device(self._device, "SuperPointExtractor")
        cfg = self.config.get("extractor")
        self._arith = _apply_arithmetic(cfg, self._lib)
        self._on_sat = _saturation_policy(cfg)
        path = cfg.get("weights_path") or os.environ.get("DIM_SUPERPOINT_WEIGHTS")
        if path is None and cfg.get("allow_synthetic_weights"):
            logger.warning("SuperPoint: running on seeded SYNTHETIC weights (allow_synthetic_weights) - test / benchmark use only")
        self._sd = _weights.load_superpoint_state_dict(path, allow_synthetic=bool(cfg.get("allow_synthetic_weights", False)))
        self._net_cfg = {k: cfg[k] for k in ("nms_radius", "keypoint_threshold", "max_keypoints", "remove_borders", "fix_sampling")}
        self._net: Optional[SuperPointHIP] = None
        self._net_hw = (0, 0)
        self._min_capacity = 0  # raised when a keep-all extraction overflowed the slot (see _regrow)

    def _capacity(self, H: int, W: int) -> int:
        """Slot size per image.  max_keypoints > 0: exactly that.  Keep-all mode (-1, the plugin default): a first
        guess of 4 keypoints per 8x8 cell; a call that yields more candidates is repeated with a larger slot
        (_regrow), so nothing is ever dropped — the reference returns every keypoint (SPN:183-207)."""
        mk = self._net_cfg["max_keypoints"]
        return mk if mk > 0 else max(self._min_capacity, min(4096 * 4, max(1024, (H // 8) * (W // 8) * 4)))

    def _ensure(self, H: int, W: int):
        cap = self._capacity(H, W)
        if self._net is None or H > self._net_hw[0] or W > self._net_hw[1] or cap > self._net.capacity:
            hw = (max(H, self._net_hw[0]), max(W, self._net_hw[1]))
            self._net = SuperPointHIP(self._sd, self._net_cfg, max_batch=1, max_hw=hw, capacity=cap,
                                      device=self._device, lib=self._lib, on_saturation=self._on_sat, arithmetic=self._arith)
            self._net_hw = hw

    def _ensure_batch(self, H: int, W: int, batch: int):
        """Separate resident handle for batched tile extraction (tiling.BatchedTilingMixin)."""
        key = getattr(self, "_tile_key", None)
        cap = self._capacity(H, W)
        if key is None or H > key[0] or W > key[1] or batch > key[2] or cap > self._tile_net.capacity:
            self._tile_net = SuperPointHIP(self._sd, self._net_cfg, max_batch=batch, max_hw=(H, W), capacity=cap,
                                           device=self._device, lib=self._lib, on_saturation=self._on_sat, arithmetic=self._arith)
            self._tile_key = (H, W, batch)
        return self._tile_net

    def _regrow(self, net: SuperPointHIP, batch: int) -> bool:
        """Keep-all mode: True when the last call produced more candidates than the slot holds (the library then kept
        only the first `capacity` in row-major order).  The next _ensure / _ensure_batch builds a larger handle."""
        if self._net_cfg["max_keypoints"] > 0:
            return False
        worst = int(net.candidate_counts(batch).max())
        if worst <= net.capacity:
            return False
        self._min_capacity = 1 << (worst - 1).bit_length()
        logger.warning("SuperPoint: %d keypoints exceed the slot of %d - repeating the extraction with %d", worst, net.capacity,
                       self._min_capacity)
        return True

    @torch.no_grad()
    def _extract(self, image: np.ndarray) -> dict:
        """image: float32 HxW, values 0..255 (extractor_base.py:197-202).  Returns numpy
        keypoints (N,2) float32 (x,y), scores (N,), descriptors (256,N) (SPX:126-130)."""
        image_ = _to_device(self, _frame2array(image), self._device)      # (= _frame2tensor without its synchronisation: this call synchronises below)
        if image_.shape[1] != 1:
            raise ValueError("SuperPoint expects a single-channel image")
        H, W = int(image_.shape[-2]), int(image_.shape[-1])
        self._ensure(H, W)
        img = image_.reshape(1, H, W)
        out = self._net.extract_batch_guarded(img)
        if self._regrow(self._net, 1):
            self._ensure(H, W)
            out = self._net.extract_batch_guarded(img)
        return _features_to_numpy(self, out)

    def _frame2tensor(self, image: np.ndarray, device: str = "cuda"):
        """SPX:134-146 (through page-locked staging on a GPU)."""
        return _to_device(self, _frame2array(image), device, sync=True)


def _to_device(plugin, image: np.ndarray, device, sync: bool = False):
    """image / 255.0 (float32, the reference's values) on ``device``, same shape as ``image``.  The copy leaves a REUSED page-locked buffer
    asynchronously: ``sync=True`` (the public _frame2tensor hooks) waits for it, so that a second call cannot overwrite the source of a copy
    still in flight; _extract keeps the asynchronous form and synchronises before it returns."""
    if str(getattr(device, "type", device)).startswith("cuda"):
        st = plugin.__dict__.setdefault("_staging", _PinnedStaging())
        t = st.upload_scaled(image, device)
        if sync:
            torch.cuda.current_stream(t.device).synchronize()
        return t
    return torch.from_numpy(np.ascontiguousarray(image / 255.0, dtype=np.float32)).to(device)


def _features_to_numpy(plugin, out) -> dict:
    """(kpts [1,cap,2], scores [1,cap], desc [1,cap,D], n [1]) of a batch-1 library call -> the reference's numpy dict: keypoints (N,2), scores
    (N,), descriptors (D,N) (SPX:126-130, ALX:57-61).  On a GPU: one pinned buffer, one synchronisation; the descriptors come back as the
    transposed VIEW of an owned (N,D) array — (D,N) to every consumer, and featuresDict2Lightglue's (N,D) costs nothing."""
    kp, sc, de, n = out
    if kp.device.type == "cuda":
        st = plugin.__dict__.setdefault("_staging", _PinnedStaging())
        kp_h, sc_h, de_h, n_h = st.download([kp[0], sc[0], de[0], n[:1]], kp.device)
    else:
        kp_h, sc_h, de_h, n_h = kp[0].numpy().copy(), sc[0].numpy().copy(), de[0].numpy().copy(), n[:1].numpy()
    k = int(n_h[0])
    return {"keypoints": kp_h[:k], "scores": sc_h[:k], "descriptors": de_h[:k].T}


class AlikedExtractor(BatchedTilingMixin, _ExtractorBase):
    """extractors/aliked.py:10 — ALIKED on the gfx950 library (train-mode BatchNorm, Q7; scores are
    the dispersities, Q8 — both reproduced inside the library)."""

    _default_conf = {  # ALX:22-29 (the "name:" key with the stray colon is the reference's)
        "name:": "aliked",
        "model": "aliked-n16rot",
        "device": "cuda",
        "max_num_keypoints": 4000,
        "detection_threshold": 0.2,
        "nms_radius": 2,
    }
    required_inputs = []
    grayscale = False
    as_float = True
    descriptor_size = 128
    features_as_half = True

    def __init__(self, config):
        super().__init__(config)
        self._lib = capi.load()
        self._device = _resolve_device(self._device, "AlikedExtractor")
        cfg = self.config.get("extractor")
        # ALIKED(**cfg) reads conf.model_name (default "aliked-n16", ALN:562-567); DIM's "model" key is ignored by the net
        self._net_cfg = {"model_name": cfg.get("model_name", "aliked-n16"), "max_num_keypoints": cfg["max_num_keypoints"],
                         "detection_threshold": cfg["detection_threshold"], "nms_radius": cfg["nms_radius"]}
        path = cfg.get("weights_path") or os.environ.get("DIM_ALIKED_WEIGHTS")
        if path is None and cfg.get("allow_synthetic_weights"):
            logger.warning("ALIKED: running on seeded SYNTHETIC weights (allow_synthetic_weights) - test / benchmark use only")
        self._sd = _weights.load_aliked_state_dict(path, model_name=self._net_cfg["model_name"],
                                                   allow_synthetic=bool(cfg.get("allow_synthetic_weights", False)))
        self.descriptor_size = int(_weights.ALIKED_CFGS[self._net_cfg["model_name"]][4])   # 128; aliked-t16: 64 (the reference's class attribute says 128 for all)
        self._net: Optional[AlikedHIP] = None
        self._net_hw = (0, 0)

    def _ensure(self, H: int, W: int):
        if self._net is None or H > self._net_hw[0] or W > self._net_hw[1]:
            hw = (max(H, self._net_hw[0]), max(W, self._net_hw[1]))
            mk = self._net_cfg["max_num_keypoints"]
            self._net = AlikedHIP(self._sd, self._net_cfg, max_batch=1, max_hw=hw, capacity=mk if mk > 0 else None,
                                  device=self._device, lib=self._lib)
            self._net_hw = hw

    def _ensure_batch(self, H: int, W: int, batch: int):
        key = getattr(self, "_tile_key", None)
        if key is None or H > key[0] or W > key[1] or batch > key[2]:
            mk = self._net_cfg["max_num_keypoints"]
            self._tile_net = AlikedHIP(self._sd, self._net_cfg, max_batch=batch, max_hw=(H, W), capacity=mk if mk > 0 else None,
                                       device=self._device, lib=self._lib)
            self._tile_key = (H, W, batch)
        return self._tile_net

    @torch.no_grad()
    def _extract(self, image: np.ndarray) -> dict:
        """image: float32 HxWx3 RGB (or HxW), 0..255.  Returns numpy keypoints (N,2), descriptors
        (128,N) (ALX:57-58 transposes), scores (N,) (ALX:60-61 renames keypoint_scores)."""
        # the library reads HWC: the reference's CHW tensor (ALX:66-78, _frame2tensor below) would be transposed there and back
        a = image if image.ndim == 3 else image[..., None]
        self._ensure(a.shape[0], a.shape[1])
        return _features_to_numpy(self, self._net.extract_batch_guarded(_to_device(self, a, self._device)[None].contiguous()))

    def _frame2tensor(self, image: np.ndarray, device: str = "cuda"):
        """ALX:66-78 (through page-locked staging on a GPU)."""
        return _to_device(self, _frame2array(image), device, sync=True)


def featuresDict2Lightglue(feats: dict) -> dict:
    """matchers/lightglue.py:8-66 up to (not including) the tensor conversion: unwrap, fix the
    descriptor layout by the keypoint count ((D,N) -> (N,D)), drop path keys."""
    feats = {k: v[0] if isinstance(v, (list, tuple)) else v for k, v in feats.items()}
    if "keypoints" not in feats or "descriptors" not in feats:
        raise KeyError("features must contain 'keypoints' and 'descriptors'")
    kpts, desc = np.asarray(feats["keypoints"]), np.asarray(feats["descriptors"])
    if kpts.ndim != 2 or kpts.shape[1] != 2:
        raise ValueError(f"Invalid keypoints shape: {kpts.shape}")
    N = kpts.shape[0]
    if desc.ndim != 2:
        raise ValueError(f"Invalid descriptors shape: {desc.shape}")
    if desc.shape[1] == N and desc.shape[0] != N:
        desc = desc.T
    elif desc.shape[0] == N:
        pass
    else:
        raise ValueError(f"Descriptor / keypoint mismatch: descriptors={desc.shape}, keypoints={kpts.shape}")
    out = dict(feats)
    out["keypoints"], out["descriptors"] = kpts, desc
    out.pop("feature_path", None)
    out.pop("im_path", None)
    return out


class LightGlueMatcher(BatchedTileMatchingMixin, _MatcherBase):
    """matchers/lightglue.py:77 — LightGlue on the gfx950 library."""

    _default_conf = {
        "flash": True,   # accepted for compatibility; attention here is always the flash-attention kernel of lg_attn_x6.hip (fp16x3 / bf16x6 split arithmetic, lg_attn.hip in fp32 MFMA mode)
        "mp": False,
        "depth_confidence": 0.95,
        "width_confidence": 0.99,
        "filter_threshold": 0.1,
        "weights": None,
        # LGN:318-323,606-610: below this many keypoints the reference does not prune.  "auto" = what the reference
        # uses on a GPU (1536 with flash attention, 1024 without); -1 = its CPU behaviour (always prune), which is
        # what the CPU-generated parity goldens need.
        "pruning_min_kpts": "auto",
    }
    required_inputs = []
    min_matches = 20
    max_feat_no_tiling = 200000
    _input_dims = {"superpoint": 256, "disk": 128, "aliked": 128, "sift": 128}  # LGN:331-349

    def __init__(self, config, local_features="superpoint") -> None:
        self._localfeatures = local_features
        super().__init__(config)
        self._lib = capi.load()
        self._device = _resolve_device(self._device, "LightGlueMatcher")
        cfg = {**self._default_conf, **self.config.get("matcher", {})}
        self._arith = _apply_arithmetic(cfg, self._lib)
        self._on_sat = _saturation_policy(cfg)
        if cfg.get("mp"):
            logger.warning("LightGlue: mixed precision ('mp') is not implemented on the MI355X path; running fp32")
        self._conf = {k: cfg[k] for k in ("depth_confidence", "width_confidence", "filter_threshold")}
        self._conf["n_layers"] = int(cfg.get("n_layers", 9))
        pm = cfg.get("pruning_min_kpts", "auto")
        self._conf["pruning_min_kpts"] = (1536 if cfg.get("flash", True) else 1024) if pm == "auto" else int(pm)
        path = cfg.get("weights_path") or os.environ.get("DIM_LIGHTGLUE_WEIGHTS")
        in_dim = self._input_dims.get(local_features, 256)
        if path is None and cfg.get("allow_synthetic_weights"):
            logger.warning("LightGlue: running on seeded SYNTHETIC weights (allow_synthetic_weights) - test / benchmark use only")
        self._sd = _weights.load_lightglue_state_dict(path, input_dim=in_dim, n_layers=self._conf["n_layers"],
                                                      allow_synthetic=bool(cfg.get("allow_synthetic_weights", False)))
        self._net: Optional[LightGlueHIP] = None
        self._net_n = 0
        if self._localfeatures == "disk":
            self.max_feat_no_tiling = 50000

    def _ensure(self, n: int):
        if self._net is None or n > self._net_n:
            dev = self._device if isinstance(self._device, (str, torch.device)) else "cuda"
            # a handle is rebuilt when a pair has more keypoints than it holds (~0.5 - 1 s: weight splitting, uploads, allocations).  Since round 6 an oversized
            # handle costs nothing measurable (launch shapes follow the pair, not the handle: 2048 keypoints on a 4096 / 8192-row handle 1.464 / 1.485 vs 1.461 ms,
            # scripts/gpu_lg_capacity_cost.py) and a 4096-row handle is ~0.2 GB, so on a GPU the first one already holds 4096 keypoints per image
            floor = 4096 if str(getattr(dev, "type", dev)).startswith("cuda") else 256
            self._net_n = max(floor, 1 << (max(n, 1) - 1).bit_length())
            self._net = LightGlueHIP(self._sd, self._conf, max_pairs=1, max_kpts=self._net_n, device=dev, lib=self._lib,
                                     on_saturation=self._on_sat, arithmetic=self._arith)

    def _ensure_pairs(self, n: int, pairs: int):
        """Batched instance for tile-pair matching (tile_matching.BatchedTileMatchingMixin)."""
        cur = getattr(self, "_net_b", None)
        if cur is None or n > self._net_b_n or pairs > self._net_b_p:
            self._net_b_n = max(256, 1 << (max(n, 1) - 1).bit_length(), getattr(self, "_net_b_n", 0))
            self._net_b_p = max(pairs, getattr(self, "_net_b_p", 0))
            dev = self._device if isinstance(self._device, (str, torch.device)) else "cuda"
            self._net_b = LightGlueHIP(self._sd, self._conf, max_pairs=self._net_b_p, max_kpts=self._net_b_n, device=dev, lib=self._lib,
                                       on_saturation=self._on_sat, arithmetic=self._arith)
        return self._net_b

    @torch.no_grad()
    def _match_pairs(self, feats0: dict, feats1: dict) -> np.ndarray:
        """feats: numpy dicts as read from features.h5 (keypoints (N,2), descriptors (D,N) or (N,D),
        image_size (2,) = (H,W), + ignored keys).  Returns (S,2) int64 index pairs (LGX:102-125)."""
        f0, f1 = featuresDict2Lightglue(feats0), featuresDict2Lightglue(feats1)
        self._ensure(max(f0["keypoints"].shape[0], f1["keypoints"].shape[0]))

        def img(f):
            d = {"keypoints": torch.as_tensor(f["keypoints"], dtype=torch.float32)[None],
                 "descriptors": torch.as_tensor(f["descriptors"], dtype=torch.float32)[None]}
            if "image_size" in f:
                d["image_size"] = torch.as_tensor(np.asarray(f["image_size"]), dtype=torch.float32).reshape(1, 2)
            else:  # LGN:26-27: size inferred from the keypoint extent
                k = d["keypoints"][0]
                d["image_size"] = (1 + k.max(0).values - k.min(0).values).reshape(1, 2) if k.numel() else torch.ones(1, 2)
            return d

        dev = torch.device(self._net.device)
        if dev.type == "cuda":
            return self._match_pairs_staged(f0, f1, dev)
        res = self._net({"image0": img(f0), "image1": img(f1)})
        return res["matches"][0].cpu().numpy()

    def _match_pairs_staged(self, f0: dict, f1: dict, dev) -> np.ndarray:
        """The same call with ONE host-to-device and ONE device-to-host transfer (round 5) of the arrays AS THE CALLER HOLDS THEM (round 6).
        The reference's loop hands ``_match_pairs`` what features.h5 holds: float16 arrays, SuperPoint / ALIKED descriptors as (D, N)
        (extractors/extractor_base.py:56-99, extractors/superpoint.py:121-127).  Converting them on the host — numpy's half -> float cast through a
        transposed view — costs 1.2 - 1.6 ms per image, more than the whole match on the device; here the raw bytes of both images' keypoints and
        descriptors go straight into one page-locked buffer (a memcpy), up in one copy, and ``dim_lg_stage_features`` builds the fp32 (N, D)
        feature table on the device (transpose through LDS; fp16 -> fp32 is exact, so the table equals the host conversion's).  The match count
        and the (S, 2) index table share one device buffer and come back in one copy, enqueued behind the match so that the range guard's
        synchronisation covers it."""
        net = self._net
        D = net.input_dim

        def raw(f):
            k, d = f["keypoints"], f["descriptors"]
            if k.dtype not in (np.float16, np.float32):
                k = k.astype(np.float32)
            k = np.ascontiguousarray(k)
            if d.dtype not in (np.float16, np.float32):
                d = d.astype(np.float32)
            dn = 0
            if not d.flags.c_contiguous:
                if d.T.flags.c_contiguous:    # featuresDict2Lightglue's transposed view of a (D, N) array: keep the array
                    d, dn = d.T, 1
                else:
                    d = np.ascontiguousarray(d)
            return k, d, dn

        def size_of(f, k):   # LGN:26-27: size inferred from the keypoint extent when the features carry none
            if "image_size" in f:
                return np.asarray(f["image_size"], dtype=np.float32).reshape(2)
            k = np.asarray(k, dtype=np.float32)
            return (1 + k.max(0) - k.min(0)).astype(np.float32) if k.size else np.ones(2, np.float32)

        (k0, d0, dn0), (k1, d1, dn1) = raw(f0), raw(f1)
        m, n = k0.shape[0], k1.shape[0]
        if (m and (d0.shape[0] if dn0 else d0.shape[1]) != D) or (n and (d1.shape[0] if dn1 else d1.shape[1]) != D):
            raise ValueError(f"descriptor dimension {d0.shape} / {d1.shape} does not match the matcher's input_dim {D}")
        cap = max(m, n, 1)
        # raw staging: [sizes 4 f32 | counts 2 i32 | pad] then the four arrays at 256-byte boundaries
        offs, cur = [], 256
        for arr in (k0, d0, k1, d1):
            offs.append(cur)
            cur += (arr.nbytes + 255) & ~255
        st = self.__dict__.setdefault("_staging", _PinnedStaging())
        pin = st.get("lg_in", cur)[:cur]
        h = pin.numpy()
        h[:16].view(np.float32)[0:2] = size_of(f0, k0)
        h[:16].view(np.float32)[2:4] = size_of(f1, k1)
        h[16:24].view(np.int32)[:] = (m, n)
        for arr, o in zip((k0, d0, k1, d1), offs):
            if arr.nbytes:
                h[o:o + arr.nbytes] = arr.reshape(-1).view(np.uint8)
        lean = self.__dict__.get("_lean")
        NK = net.nk
        tab_floats = 2 * cap * (2 + D)
        if lean is None or lean["net"] is not net or lean["raw"].numel() < cur or lean["tab"].numel() < tab_floats:
            flat = torch.zeros(2 + NK * 2, dtype=torch.int64, device=dev)       # [n_matches (int32) | pad | matches NK x 2]
            out = {"matches": flat[2:].view(1, NK, 2), "scores": torch.zeros(1, NK, dtype=torch.float32, device=dev),
                   "n_matches": flat[:1].view(torch.int32)[:1], "matches01": torch.zeros(1, 2, NK, dtype=torch.int32, device=dev),
                   "mscores01": torch.zeros(1, 2, NK, dtype=torch.float32, device=dev), "stop": torch.zeros(1, dtype=torch.int32, device=dev),
                   "prune01": torch.zeros(1, 2, NK, dtype=torch.int32, device=dev)}
            lean = {"net": net, "raw": torch.empty(max(cur, 256 + 4 * 2 * NK * (2 + D) + 1024), dtype=torch.uint8, device=dev),
                    "tab": torch.empty(max(tab_floats, 2 * NK * (2 + D)), dtype=torch.float32, device=dev), "flat": flat, "out": out}
            self.__dict__["_lean"] = lean
        rawd, tab = lean["raw"], lean["tab"]
        pout = st.get("lg_out", (2 + NK * 2) * 8)[: (2 + NK * 2) * 8].view(torch.int64)
        base = rawd.data_ptr()
        descr = [capi.LgRawFeatures(base + offs[0], base + offs[1], m, int(k0.dtype == np.float16), int(d0.dtype == np.float16), dn0),
                 capi.LgRawFeatures(base + offs[2], base + offs[3], n, int(k1.dtype == np.float16), int(d1.dtype == np.float16), dn1)]
        kt, dt = tab[: 4 * cap].view(2, cap, 2), tab[4 * cap: 4 * cap + 2 * cap * D].view(2, cap, D)
        sizes, counts = rawd[:16].view(torch.float32).view(2, 2), rawd[16:24].view(torch.int32)

        def run():
            rawd[:cur].copy_(pin, non_blocking=True)
            capi.check(net.lib, net.lib.dim_lg_stage_features(ctypes.byref(descr[0]), ctypes.byref(descr[1]), int(cap), int(D), capi.ptr(kt), capi.ptr(dt), net._stream()))
            net.match_batch(kt, dt, counts, sizes, n_pairs=1, out=lean["out"])
            pout.copy_(lean["flat"], non_blocking=True)

        with net._ctx():
            capi.run_guarded(net.lib, net._stream(), run, "LightGlue", net.on_saturation, logger, handle=net._h, arithmetic=net.arithmetic)
        torch.cuda.current_stream(dev).synchronize()
        res = pout.numpy()
        S = int(res[:1].view(np.int32)[0])
        return res[2:2 + 2 * S].reshape(S, 2).copy()
