"""Where the 16-tile extraction of a 6000x4000 RGB float32 image (config 5, ALIKED) spends its wall time: H2D of the caller's array,
tile gather + network, D2H of the tables, host merge (EB:330-390)."""
import importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
plugins = importlib.import_module('deep-image-matching_amd.plugins'); tiling = importlib.import_module('deep-image-matching_amd.tiling')
general = {"tile_size": (1500, 1000), "tile_overlap": 0, "tile_preselection_size": 1024, "min_matches_per_tile": 5, "quality": "HIGH", "allow_synthetic_weights": True}
ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 4000,
                                                                 "detection_threshold": 0.2, "nms_radius": 3, "allow_synthetic_weights": True}})
rng = np.random.default_rng(0)
u8 = rng.integers(0, 256, (4000, 6000, 3), dtype=np.uint8)
img = u8.astype(np.float32)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return r, round((time.perf_counter() - t0) / reps * 1e3, 2)
res = {}
_, res["h2d_f32_288MB_pageable_ms"] = t(lambda: torch.from_numpy(img).to("cuda"))
_, res["h2d_u8_72MB_pageable_ms"] = t(lambda: torch.from_numpy(u8).to("cuda"))
pin = torch.empty(img.shape, dtype=torch.float32).pin_memory()
_, res["host_copy_to_pinned_1thread_ms"] = t(lambda: pin.numpy().__setitem__(slice(None), img))
_, res["h2d_f32_pinned_ms"] = t(lambda: pin.to("cuda", non_blocking=True))
f, res["extract_by_tile_total_ms"] = t(lambda: ex._extract_by_tile(img))
net = ex._ensure_batch(1000, 1500, ex.tile_batch)
tiles = torch.rand(16, 1000, 1500, 3, device="cuda")
o, res["network_16_tiles_ms"] = t(lambda: net.extract_batch(tiles))
_, res["d2h_tables_ms"] = t(lambda: [x.cpu().numpy() for x in o])
kp, sc, de, n = [x.cpu().numpy() for x in o]
per_tile = {i: {"keypoints": kp[i, :n[i]].copy(), "scores": sc[i, :n[i]].copy(), "descriptors": de[i, :n[i]].T.copy()} for i in range(16)}
origins = {r * 4 + c: (c * 1500, r * 1000) for r in range(4) for c in range(4)}
t0 = time.perf_counter(); [kp[i, :n[i]].copy() or de[i, :n[i]].T.copy() for i in range(0)]; 
t0 = time.perf_counter()
for _ in range(3):
    pt = {i: {"keypoints": kp[i, :n[i]].copy(), "scores": sc[i, :n[i]].copy(), "descriptors": de[i, :n[i]].T.copy()} for i in range(16)}
res["host_untable_transpose_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
t0 = time.perf_counter()
for _ in range(3): tiling.merge_tile_features(per_tile, origins, img.shape, 128, True)
res["host_merge_unique_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
res["keypoints"] = int(f["keypoints"].shape[0])
print(json.dumps(res))
