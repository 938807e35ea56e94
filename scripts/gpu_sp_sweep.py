"""MI355X: SuperPoint through the batched entry (resident tensors, HIP events): ms per image over the image size at one image per call, and over the images per
call at 1024 x 1024 — looking for cliffs where kernel selections hand over."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
sp = importlib.import_module('deep-image-matching_amd.superpoint_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sd = weights.synthetic_superpoint_state_dict(0)
cfg = {"nms_radius": 3, "keypoint_threshold": 0.005, "max_keypoints": 2048}
res = {"one_image": {}, "batch_1024": {}}
g = torch.Generator().manual_seed(1)
for S in (256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096):
    net = sp.SuperPointHIP(sd, cfg, max_batch=1, max_hw=(S, S))
    img = torch.rand(1, S, S, generator=g).cuda()
    o = [None]
    def f(): o[0] = net.extract_batch(img, out=o[0])
    ms = timeit(f)
    res["one_image"][S] = {"ms": round(ms, 4), "ns_per_px": round(ms * 1e6 / S / S, 3)}
    del net, o; torch.cuda.empty_cache()
for B in (1, 2, 3, 4, 6, 8, 16, 32, 100):
    net = sp.SuperPointHIP(sd, cfg, max_batch=B, max_hw=(1024, 1024))
    img = torch.rand(B, 1024, 1024, generator=g).cuda()
    o = [None]
    def f(): o[0] = net.extract_batch(img, out=o[0])
    ms = timeit(f, 5)
    res["batch_1024"][B] = {"ms_per_image": round(ms / B, 4)}
    del net, o; torch.cuda.empty_cache()
print(json.dumps(res))
