"""Measurement behind DESIGN.md section 8 "cross attention with a shared score tile" (VERDICT r2 item 7).

LightGlue's cross block computes sim = q0 k1^T once and uses it for both directions (LGN:197-206).  attn_x6_kernel evaluates the
score tile once per direction (two flash-style passes, no score tile ever stored).  A kernel in which ONE workgroup owns a
(query block of image 0) x (key tile of image 1) score tile and feeds both directions executes 36 instead of 48 MFMAs per tile
pair, but the second direction's output cannot stay in registers across query blocks: per 32-key tile it leaves a partial record
(32 x (64 dims + running max + running sum)) that a merge pass folds over the 16 query blocks.

This script times the cross-attention launches (HIP events on the launch stream, dim_profile site DIM_PROF_LG_CROSS_ATTN) of the
bench batch (50 pairs x 2048 keypoints, 9 layers, fixed work) under dim_tune_set key 12:
  0  the product kernel;
  1  every second key tile WITHOUT its score MFMAs (-25 % of the launch's MFMAs) — the most sharing could save (results wrong,
     timing only);
  2  every query block's key range cut in 16 -> 16 partial records per row written and merged (results stay correct):
     the record volume of the shared tile for BOTH directions, through the existing split + combine path;
  3  the product kernel additionally WRITING one direction's partial records (every second key tile, 1.7 GB per launch; timing only).
Prints one JSON object; ms are per cross-attention launch (100 items)."""
import ctypes, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(str(capi.LIB_PATH.parent / 'libdim_hip_research.so')); capi.install(lib, None)   # research build: dim_tune_set keys 12-15 (timing probes / prototypes) exist only there
P, N = 50, 2048
sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
g = torch.Generator().manual_seed(0)
kt = (torch.rand(2 * P, N, 2, generator=g) * 1024).cuda()
dt = torch.nn.functional.normalize(torch.randn(2 * P, N, 256, generator=g), dim=-1).cuda()
nt = torch.full((2 * P,), N, dtype=torch.int32).cuda(); st = torch.tensor([[1024.0, 1024.0]] * (2 * P)).cuda()
res = {}
for probe in (0, 1, 2, 3, 0):
    lib.dim_tune_set(12, probe)
    net = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N)
    for _ in range(2):
        net.match_batch(kt, dt, nt, st, n_pairs=P)
    torch.cuda.synchronize()
    out = {}
    for name, site in (("cross", 14), ("self", 13)):
        capi.check(lib, lib.dim_profile_start(ctypes.c_ulonglong(1 << site)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net.match_batch(kt, dt, nt, st, n_pairs=P)
        e1.record(); torch.cuda.synchronize()
        ms, n = ctypes.c_double(), ctypes.c_int()
        capi.check(lib, lib.dim_profile_stop(ctypes.byref(ms), ctypes.byref(n)))
        out[name + "_ms_per_launch"] = round(ms.value / max(1, n.value), 4); out[name + "_launches"] = n.value
        out["match_ms_per_batch"] = round(e0.elapsed_time(e1) / 3, 3)
    res[f"probe{probe}" + ("_again" if probe == 0 and "probe0" in res else "")] = out
    del net
lib.dim_tune_set(12, 0)
b = res["probe0"]["cross_ms_per_launch"]
res["summary"] = {"max_saving_ms_per_launch": round(b - res["probe1"]["cross_ms_per_launch"], 4),
                  "split16_write_and_merge_both_directions_ms": round(res["probe2"]["cross_ms_per_launch"] - b, 4),
                  "record_writes_one_direction_ms": round(res["probe3"]["cross_ms_per_launch"] - b, 4)}
print(json.dumps(res, indent=1))
