"""MI355X: two more one-call sweeps looking for cliffs — SuperPoint on one 1024 x 1024 noise image over max_keypoints (the selection switches kernels above 4096),
and LightGlue on one pair with 128-d descriptors (input_proj) against 256-d."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
sp = importlib.import_module('deep-image-matching_amd.superpoint_hip'); lg = importlib.import_module('deep-image-matching_amd.lightglue_hip')
weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {"superpoint_max_keypoints": {}, "lightglue_input_dim": {}}
g = torch.Generator().manual_seed(1)
img = torch.rand(1, 1024, 1024, generator=g).cuda()
sd = weights.synthetic_superpoint_state_dict(0)
for k in (1024, 2048, 4096, 4097, 8000, 16000, -1):
    net = sp.SuperPointHIP(sd, {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": k}, max_batch=1, max_hw=(1024, 1024))
    o = [None]
    def f(): o[0] = net.extract_batch(img, out=o[0])
    res["superpoint_max_keypoints"][k] = round(timeit(f), 4)
    del net, o; torch.cuda.empty_cache()
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
for D in (256, 128, 64):
    sdl = weights.synthetic_lightglue_state_dict(0, D, gain=2.0)
    kt = (torch.rand(2, 2048, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, 2048, D, generator=g), dim=-1).cuda()
    nt = torch.full((2,), 2048, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
    m = lg.LightGlueHIP(sdl, conf, max_pairs=1, max_kpts=2048)
    q = [None]
    def f2(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
    res["lightglue_input_dim"][D] = round(timeit(f2), 4)
    del m, q; torch.cuda.empty_cache()
print(json.dumps(res))
