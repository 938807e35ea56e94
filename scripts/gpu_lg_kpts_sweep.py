"""MI355X: ONE pair per call, fixed work, n x n keypoints for n = 256 ... 4096 (handle sized for n): ms per pair — is there a cliff where the one-pair kernel
selections (32 x 128 GEMM blocks, K | V images from the projection, key-split attention) hand over to the general ones?"""
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
for N in (256, 512, 1024, 1536, 2048, 2304, 2560, 3072, 4096, 6144, 8192):
    kt = (torch.rand(2, N, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, N, 256, generator=g), dim=-1).cuda()
    nt = torch.full((2,), N, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
    m = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=N)
    q = [None]
    def f(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    ms = timeit(f)
    res[N] = {"ms_per_pair": round(ms, 4), "us_per_kpt": round(ms * 1e3 / N, 3), "ns_per_kpt2": round(ms * 1e6 / N / N, 4)}
    del m, q; torch.cuda.empty_cache()
print(json.dumps(res))
