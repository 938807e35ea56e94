"""GPU A/B: conv3x3_x6 workgroup-tile variants (dim_tune_set key 2: bit 4 = 16-row tiles, bit 5 = 3 kernel rows of weights per
stage) on the SuperPoint extraction of the bench batch (100 images 1024^2, sustained), results must be bit-identical."""
import ctypes, importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load()
sp = importlib.import_module('deep-image-matching_amd.superpoint_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
B = int(os.environ.get("B", 100)); ITERS = int(os.environ.get("ITERS", 12))
net = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=B, max_hw=(1024, 1024), capacity=2048)
imgs = torch.rand(B, 1024, 1024, device='cuda')
res, ref = {}, None
variants = [1, 17]
for rnd in range(3):
    for v in variants:
        lib.dim_tune_set(2, v)
        out = net.extract_batch(imgs); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS): out = net.extract_batch(imgs)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / ITERS
        res.setdefault(str(v), []).append(round(ms / B * 1000, 1))   # us per image
        sig = (out[0].clone(), out[2].clone())
        if ref is None: ref = sig
        else: assert torch.equal(ref[0], sig[0]) and torch.equal(ref[1], sig[1]), ("variant changes results", v)
lib.dim_tune_set(2, 17)
print(json.dumps({"us_per_image": res, "batch": B}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"us_per_image": res, "batch": B}, open("gpurun_out/conv_tiles.json", "w"))
