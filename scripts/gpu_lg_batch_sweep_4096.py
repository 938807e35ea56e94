import importlib, json, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sd = weights.synthetic_lightglue_state_dict(0, 128, gain=2.0)
g = torch.Generator().manual_seed(0)
res = {}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for adaptive in (0, 1):
    conf = {"depth_confidence": 0.95 if adaptive else -1, "width_confidence": 0.99 if adaptive else -1, "filter_threshold": 0.1}
    for P in (8, 16, 24, 28, 32, 48):
        kt = (torch.rand(2 * P, N, 2, generator=g) * 1500).cuda(); dt = torch.nn.functional.normalize(torch.randn(2 * P, N, 128, generator=g), dim=-1).cuda()
        nt = torch.full((2 * P,), N, dtype=torch.int32, device='cuda'); st = torch.full((2 * P, 2), 1500.0, device='cuda')
        m = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N)
        q = [None]
        def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
        ms = timeit(f)
        res[f"adaptive{adaptive}_P{P}"] = round(ms / P, 4)
        del m, q; torch.cuda.empty_cache()
print(json.dumps(res))
