"""Phase timers of the fused feed-forward kernel (gemm_x6_ffn_fused_kernel) on the bench shape: s_memtime stamps inside an INSTRUMENTED build
(deep-image-matching_amd/lib/libdim_hip_ffntime.so = build.build_variant("ffntime", ["-DDIM_FFN_TIMERS", "-DDIM_RESEARCH"]) — -DDIM_RESEARCH only when a prototype loop (argument = dim_tune_set key 14 value) is timed), summed over every wave.
Build the variant in the build container first (it is not part of build()), then on the GPU box:
python scripts/gpu_ffn_phases.py [14=VALUE]  ->  one JSON line: average ns per wave and phase, and each phase's share."""
import ctypes, importlib, json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
capi = importlib.import_module("deep-image-matching_amd.capi")
lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
weights = importlib.import_module("deep-image-matching_amd.weights")
lib = capi.load(str(ROOT / "deep-image-matching_amd" / "lib" / "libdim_hip_ffntime.so"))
P = 50
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=P, max_kpts=2048, device="cuda", lib=lib)
g = torch.Generator().manual_seed(0)
kp = (torch.rand(2 * P, 2048, 2, generator=g) * 1024).cuda()
de = torch.nn.functional.normalize(torch.randn(2 * P, 2048, 256, generator=g), dim=-1).cuda()
n = torch.full((2 * P,), 2048, dtype=torch.int32).cuda()
sz = torch.full((2 * P, 2), 1024.0).cuda()
if len(sys.argv) > 1:
    lib.dim_tune_set(14, int(sys.argv[1]))      # e.g. 34: the rolling-request prototype
mat.match_batch(kp, de, n, sz); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 24)()
lib.dim_ffn_phase_read(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); mat.match_batch(kp, de, n, sz); e1.record(); torch.cuda.synchronize()
lib.dim_ffn_phase_read(buf, 1)
names = {13: "prologue (to the first chunk)", 0: "ffn.0 loop: wait for the chunk + split + LDS store", 1: "ffn.0 loop: barrier after the store",
         2: "ffn.0 loop: fragment requests + MFMA steps", 3: "ffn.0 loop: barrier after the MFMAs", 4: "scale + bias + LayerNorm statistics (2 exchanges)",
         5: "normalise + GELU + guard + split", 6: "ffn.3: MFMA steps (4 rounds)", 7: "ffn.3: residual requests + exchange stores", 8: "ffn.3: exchange barrier",
         9: "ffn.3: reduce + scale + bias + residual + store", 10: "ffn.3: barrier before the next round"}
waves = max(1, buf[17])
ns_per_tick = buf[18] * 10.0 / max(1, buf[16])          # s_memtime ticks calibrated against s_memrealtime (100 MHz)
tot = buf[16] / waves * ns_per_tick
t_inst = e0.elapsed_time(e1)
# the same call through the product library: what the instrumentation costs
lib0 = capi.load()
mat0 = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=P, max_kpts=2048, device="cuda", lib=lib0)
mat0.match_batch(kp, de, n, sz); torch.cuda.synchronize()
e0.record(); mat0.match_batch(kp, de, n, sz); e1.record(); torch.cuda.synchronize()
row = {"tune_14": int(sys.argv[1]) if len(sys.argv) > 1 else 32, "waves_reporting": int(buf[17]), "ns_per_s_memtime_tick": round(ns_per_tick, 4), "us_per_wave": round(tot / 1e3, 2),
       "lightglue_call_ms_instrumented": round(t_inst, 2), "lightglue_call_ms_product": round(e0.elapsed_time(e1), 2),
       "phases_us": {v: round(buf[k] / waves * ns_per_tick / 1e3, 2) for k, v in names.items()},
       "phases_share": {v: round(buf[k] * ns_per_tick / waves / tot, 3) for k, v in names.items()}}
row["unaccounted_share"] = round(1.0 - sum(row["phases_share"].values()), 3)
print(json.dumps(row))
