import importlib, json, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
g = torch.Generator().manual_seed(0)
res = {}
for adaptive in (0, 1):
    conf = {"depth_confidence": 0.95 if adaptive else -1, "width_confidence": 0.99 if adaptive else -1, "filter_threshold": 0.0}
    for N in (500, 1000, 2048):
        for cap in (max(256, 1 << (N - 1).bit_length()), 4096, 8192):
            kt = (torch.rand(2, N, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, N, 256, generator=g), dim=-1).cuda()
            nt = torch.full((2,), N, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
            m = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap)
            q = [None]
            def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
            res[f"adaptive{adaptive}_{N}_of_{cap}"] = round(timeit(f), 4)
            del m, q; torch.cuda.empty_cache()
print(json.dumps(res))
