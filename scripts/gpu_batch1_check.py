"""GPU: the batch-1 calls of the plugin hooks on resident tensors (HIP events): SuperPoint on one 1024 x 1024 image, LightGlue on one 2048 x 2048 pair
(9 layers, fixed work), the latter with the key range of its attention launches cut into 4 (product), 8 or 16 parts (research library,
dim_tune_set(12, 8 / 16), set before the matcher allocates its partial-result scratch) and with two instead of one key tiles of prefetch distance (12 = 22)."""
import ctypes, importlib, json, os, sys
import numpy as np, torch
torch.set_num_threads(min(16, os.cpu_count() or 16))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module('deep-image-matching_amd.capi')
build = importlib.import_module('deep-image-matching_amd.build')
lib = capi.load(str(build.LIBDIR / "libdim_hip_research.so"))
capi.install(lib, None)
plugins = importlib.import_module('deep-image-matching_amd.plugins')
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(77)


def t_ms(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


res = {}
ex = plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", "nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048,
                                                               "remove_borders": 4, "allow_synthetic_weights": True}})
imgs = [(torch.rand(1024, 1024, generator=g) * 255).numpy().astype(np.float32) for _ in range(2)]
feats = []
for im in imgs:
    f = ex._extract(im); f["image_size"] = np.array([1024, 1024], dtype=np.int32); feats.append(f)
net = ex._net
img_d = torch.from_numpy(imgs[0] / 255.0).to(dev)[None].contiguous()
out_sp = net.extract_batch(img_d)
res["superpoint_batch1_ms"] = t_ms(lambda: net.extract_batch(img_d, out=out_sp))
kp, sc, de, n = out_sp
kt = torch.stack([kp[0], kp[0]]).contiguous(); dt_ = torch.stack([de[0], de[0]]).contiguous()
nt = torch.stack([n[0], n[0]]).contiguous(); st = torch.full((2, 2), 1024.0, device=dev)
ref = None
for splits in (0, 22, 0, 22, 8):
    assert lib.dim_tune_set(12, splits) == 0, lib.dim_last_error()
    mt = plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1,
                                                               "allow_synthetic_weights": True}}, local_features="superpoint")
    mt._match_pairs(feats[0], feats[1])
    lgn = mt._net
    out = lgn.match_batch(kt, dt_, nt, st, n_pairs=1)
    torch.cuda.synchronize()
    nm = int(out["n_matches"][0]); m = out["matches"][0, :nm].cpu()
    if ref is None:
        ref = m
    res.setdefault("lightglue_batch1", []).append({"knob_12": splits, "what": {0: "product: 4 key parts, one key tile ahead", 22: "4 key parts, two key tiles ahead (prototype)", 8: "8 key parts", 16: "16 key parts"}[splits], "ms_per_pair": t_ms(lambda: lgn.match_batch(kt, dt_, nt, st, n_pairs=1, out=out)),
                                                   "matches": nm, "same_matches_as_4_splits": bool(m.shape == ref.shape and torch.equal(m, ref))})
    del mt, lgn
    torch.cuda.empty_cache()
lib.dim_tune_set(12, 0)
print(json.dumps(res))
