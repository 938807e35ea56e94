"""MI355X: ONE pair per call, fixed work, n live keypoints in a table of n rows on a handle of the next power of two (what LightGlueMatcher._ensure allocates): ms per pair."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
# handle capacity = next power of two (what the plugin's _ensure allocates), live keypoints n
for N in (2100, 2304, 2560, 3072, 4096, 5000, 6144, 8192):
    cap = max(256, 1 << (N - 1).bit_length())
    kt = (torch.rand(2, N, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, N, 256, generator=g), dim=-1).cuda()   # the hook's table: max(m, n) rows per image
    nt = torch.full((2,), N, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
    m = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap)
    q = [None]
    def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    res[f"{N}_of_{cap}"] = round(timeit(f), 4)
    del m, q; torch.cuda.empty_cache()
print(json.dumps(res))
