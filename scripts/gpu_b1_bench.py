"""Batch-1 (plugin-hook shaped) timings: one 1024x1024 image through SuperPoint, one 2048x2048-keypoint pair through
LightGlue (adaptive stops off, all 9 layers)."""
import importlib, sys, json, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); sp = importlib.import_module('deep-image-matching_amd.superpoint_hip')
weights = importlib.import_module('deep-image-matching_amd.weights')
res = {}
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
net = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(0), {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048}, max_batch=1, max_hw=(1024, 1024))
img = torch.rand(1, 1024, 1024, device='cuda')
o = [None]
def f(): o[0] = net.extract_batch(img, out=o[0])
res['sp_B1_ms'] = timeit(f)
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
m = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=1, max_kpts=2048)
kt = torch.rand(2, 2048, 2, device='cuda') * 1024; dt = torch.nn.functional.normalize(torch.randn(2, 2048, 256, device='cuda'), dim=-1)
nt = torch.full((2,), 2048, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
q = [None]
def g(): q[0] = m.match_batch(kt, dt, nt, st, out=q[0])
res['lg_B1_ms'] = timeit(g)
print(json.dumps(res))
