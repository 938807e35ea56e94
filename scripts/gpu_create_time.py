import importlib, json, os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); sp = importlib.import_module('deep-image-matching_amd.superpoint_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
sd = weights.synthetic_lightglue_state_dict(0, 256)
conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}
res = {}
torch.cuda.init(); torch.zeros(1, device='cuda')
for tag, (P, N) in {"one_pair_2048": (1, 2048), "one_pair_4096": (1, 4096), "pairs50_2048": (50, 2048)}.items():
    ts = []
    for _ in range(3):
        t = time.perf_counter(); m = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N); torch.cuda.synchronize(); ts.append(round(time.perf_counter() - t, 3)); del m
    res["lightglue_" + tag] = ts
ssd = weights.synthetic_superpoint_state_dict(0)
ts = []
for _ in range(3):
    t = time.perf_counter(); n = sp.SuperPointHIP(ssd, {"nms_radius": 3, "keypoint_threshold": 0.005, "max_keypoints": 2048}, max_batch=1, max_hw=(1024, 1024)); torch.cuda.synchronize(); ts.append(round(time.perf_counter() - t, 3)); del n
res["superpoint_1024"] = ts
print(json.dumps(res))
