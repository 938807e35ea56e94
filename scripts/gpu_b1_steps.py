"""MI355X (round 6): one 2048 x 2048-keypoint pair through LightGlue (fixed work, 9 layers), batch 1 — what each of round 6's one-pair changes is worth.
Research library (key 14 = 70 brings back round 5's 64 x 128 small-problem GEMM block, 78 its 32-wide-chunk 32 x 128 successor): ms per pair with HIP events for
  round5      : 64-row GEMM blocks, separate kv_prep launch  (keys 14 = 70, 8 = 0)
  +gemm32     : the 32 x 128 block with waves 1 x 4 (32-wide chunks)
  +kc64       : ... with 64-wide K chunks (key 14 = 79: step-pipelined fragment requests)
  +chunk-ahead: ... weight fragments and activations requested a whole chunk period ahead (BSET in gemm_x6.hip)
  +kv         : the q|k|v projection writes the K | V tile images (no kv_prep launch) = the product default
and whether the match lists agree with the first setting."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); build = importlib.import_module('deep-image-matching_amd.build')
lib = capi.load(str(build.LIBDIR / "libdim_hip_research.so")); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
g = torch.Generator().manual_seed(0)
kt = (torch.rand(2, 2048, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, 2048, 256, generator=g), dim=-1).cuda()
nt = torch.full((2,), 2048, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
res = {}
ref = None
steps = (("round5", 70, 0, 3), ("+gemm32", 78, 0, 3), ("+kc64", 79, 0, 3), ("+chunk-ahead requests", 0, 0, 3), ("+kv = product", 0, 1, 3), ("round5 again", 70, 0, 3), ("product again", 0, 1, 3))
for name, k14, k8, k11 in steps:
    for k, v in ((14, k14), (8, k8), (11, k11)): assert lib.dim_tune_set(k, v) == 0, lib.dim_last_error()
    m = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, gain=2.0), conf, max_pairs=1, max_kpts=2048)
    q = [None]
    def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    res[name] = {"ms_per_pair": round(timeit(f), 4)}
    mm = q[0]["matches"][0, : int(q[0]["n_matches"][0])].cpu()
    if ref is None: ref = mm
    res[name]["matches"] = int(mm.shape[0]); res[name]["same_matches_as_round5"] = bool(mm.shape == ref.shape and torch.equal(ref, mm))
    del m
for k, v in ((14, 0), (8, 1), (11, 3)): lib.dim_tune_set(k, v)
print(json.dumps(res))
