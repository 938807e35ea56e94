"""MI355X: ALIKED (aliked-n16rot geometry, synthetic weights) through the batched entry on resident tensors: ms per tile over the tile size at one tile per call,
over the tiles per call at 1000 x 1500, and over max_num_keypoints — looking for cliffs."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
al = importlib.import_module('deep-image-matching_amd.aliked_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sd = weights.synthetic_aliked_state_dict(7)
res = {"one_tile_size": {}, "tiles_per_call_1000x1500": {}, "max_num_keypoints_2000x2000": {}}
def run(B, H, W, k, thr=0.2):
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": k, "detection_threshold": thr, "nms_radius": 2}
    net = al.AlikedHIP(sd, cfg, max_batch=B, max_hw=(H, W), capacity=k)
    imgs = torch.rand(B, H, W, 3, device='cuda')
    ms = timeit(lambda: net.extract_batch(imgs))
    n = net.extract_batch(imgs)[3].tolist()
    del net
    torch.cuda.empty_cache()
    return ms, n
for S in (256, 512, 768, 1024, 1536, 2000, 2048, 3000):
    ms, n = run(1, S, S, 4000)
    res["one_tile_size"][S] = {"ms": round(ms, 3), "ns_per_px": round(ms * 1e6 / S / S, 3), "kpts": n[0]}
for B in (1, 2, 4, 8, 16):
    ms, n = run(B, 1000, 1500, 4000)
    res["tiles_per_call_1000x1500"][B] = {"ms_per_tile": round(ms / B, 3)}
for k in (2000, 4000, 4096, 4097, 8000, 16000):
    ms, n = run(1, 2000, 2000, k, thr=-1.0)
    res["max_num_keypoints_2000x2000"][k] = {"ms": round(ms, 3), "kpts": n[0]}
print(json.dumps(res))
