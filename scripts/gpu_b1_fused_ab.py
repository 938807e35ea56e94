"""MI355X: one 2048 x 2048-keypoint pair through LightGlue (fixed work, 9 layers), batch 1 — the small-batch kernel selection (default) against the
large-batch fused kernels forced at this size (dim_tune_set 6 = 2: 128 x 256 q|k|v blocks writing the K | V images, no kv_prep pass; 11 = 4: the
one-kernel feed-forward) and the two mixed settings.  ms per pair, HIP events, + whether the match lists agree."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
capi = importlib.import_module('deep-image-matching_amd.capi')
lib = capi.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
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
for name, k6, k11 in (("default", 1, 3), ("kv_fused", 2, 3), ("ffn_fused", 1, 4), ("both", 2, 4), ("default_again", 1, 3)):
    lib.dim_tune_set(6, k6); lib.dim_tune_set(11, k11)
    m = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, gain=2.0), conf, max_pairs=1, max_kpts=2048)
    q = [None]
    def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    res[name] = round(timeit(f), 4)
    mm = q[0]["matches"][0, : int(q[0]["n_matches"][0])].cpu()
    if ref is None: ref = mm
    else: res[name + "_same_matches"] = bool(torch.equal(ref, mm))
    del m
lib.dim_tune_set(6, 1); lib.dim_tune_set(11, 3)
print(json.dumps(res))
