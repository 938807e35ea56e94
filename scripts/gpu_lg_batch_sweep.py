"""MI355X: LightGlue fixed work, 2048 x 2048 keypoints, ms per PAIR at 1 / 2 / 3 / 4 / 6 / 8 / 16 / 50 pairs per call (where do the kernel selections of the
one-pair path hand over to the batched ones?)."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
g = torch.Generator().manual_seed(0)
res = {}
for P in (1, 2, 3, 4, 6, 8, 16, 50):
    kt = (torch.rand(2 * P, 2048, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2 * P, 2048, 256, generator=g), dim=-1).cuda()
    nt = torch.full((2 * P,), 2048, dtype=torch.int32, device='cuda'); st = torch.full((2 * P, 2), 1024.0, device='cuda')
    m = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=2048)
    q = [None]
    def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    ms = timeit(f)
    res[P] = {"ms_per_call": round(ms, 3), "ms_per_pair": round(ms / P, 4)}
    del m
print(json.dumps(res))
