"""BASELINE config 5 end to end on one GPU (not a bench line): two synthetic 6000x4000 RGB images, ALIKED
(aliked-n16rot geometry, 4000 keypoints per 1500x1000 tile, 16 tiles per image) through the batched tile
extractor, tile preselection on the device (down-sampled band 1 -> SuperPoint -> LightGlue -> votes), then all
selected tile pairs through the batched LightGlue (features="aliked", 128-d).  Prints per-stage wall times."""
import importlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
plugins = importlib.import_module('deep-image-matching_amd.plugins')
tm = importlib.import_module('deep-image-matching_amd.tile_matching')

def sync_time(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return out, (time.perf_counter() - t0) / reps

general = {"tile_size": (1500, 1000), "tile_overlap": 0, "tile_preselection_size": 1024, "min_matches_per_tile": 5, "quality": "HIGH",
           "allow_synthetic_weights": True}
ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 4000,
                                                                 "detection_threshold": 0.2, "nms_radius": 3, "allow_synthetic_weights": True}})
mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", "depth_confidence": 0.95, "width_confidence": 0.99,
                                                               "filter_threshold": 0.1, "allow_synthetic_weights": True}}, local_features="aliked")
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (4000, 6000, 3), dtype=np.uint8).astype(np.float32)
img0, img1 = base, np.roll(base, (300, 500), axis=(0, 1)).copy()
res = {}
ex._extract_by_tile(img0)  # builds the resident handle
f0, res["extract_16_tiles_ms"] = sync_time(lambda: ex._extract_by_tile(img0))
f1, _ = sync_time(lambda: ex._extract_by_tile(img1))
for f, im in ((f0, img0), (f1, img1)):
    f["image_size"] = np.array(im.shape[:2], dtype=np.int32)
res["keypoints"] = [int(f0["keypoints"].shape[0]), int(f1["keypoints"].shape[0])]
band0, band1 = np.ascontiguousarray(img0[..., 0]), np.ascontiguousarray(img1[..., 0])
mt.tile_selection("a", "b", "PRESELECTION", image0=band0, image1=band1)  # builds the preselection networks
mt._tile_preselector._cache.clear()
pairs, res["preselection_cold_ms"] = sync_time(lambda: mt.tile_selection("a", "b", "PRESELECTION", image0=band0, image1=band1))
_, res["preselection_cached_ms"] = sync_time(lambda: mt.tile_selection("a", "b", "PRESELECTION", image0=band0, image1=band1))
res["tile_pairs_selected"] = len(pairs)
grid = tm.select_tile_pairs("GRID", range(16), range(16))
tm.match_tile_pairs_batched(mt._ensure_pairs, f0, f1, grid, "cuda", 8)
m, t = sync_time(lambda: tm.match_tile_pairs_batched(mt._ensure_pairs, f0, f1, grid, "cuda", 8))
res["match_16_grid_tile_pairs_ms"] = t; res["matches_grid"] = int(len(m))
if pairs:
    m2, t2 = sync_time(lambda: tm.match_tile_pairs_batched(mt._ensure_pairs, f0, f1, pairs, "cuda", 8))
    res["match_preselected_tile_pairs_ms"] = t2; res["matches_preselected"] = int(len(m2))
for k in list(res):
    if k.endswith("_ms"): res[k] = round(res[k] * 1e3, 2)
print(json.dumps(res))
