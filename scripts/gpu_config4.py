"""BASELINE configs[3] on ONE MI355X (the driver owns the 8-GPU runs): 150 synthetic 1024x1024 images -> the first
10 000 exhaustive pairs (pairs_generator.py:37-38 order) through PairMatchingPipeline.  Prints one JSON line with the
phase timings (extraction amortised over the images, one match per pair; SURVEY §8(d) "config 4 accounting").
Not the bench headline — a timing line for profiles/.

    python scripts/gpu_config4.py [--images 150] [--pairs 10000] [--mode fixed|adaptive]
"""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
PKG = "deep-image-matching_amd"

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=150)
ap.add_argument("--pairs", type=int, default=10000)
ap.add_argument("--batch", type=int, default=50)
a = ap.parse_args()

sp = importlib.import_module(PKG + ".superpoint_hip")
lg = importlib.import_module(PKG + ".lightglue_hip")
pl = importlib.import_module(PKG + ".pipeline")
weights = importlib.import_module(PKG + ".weights")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=a.batch, max_hw=(1024, 1024), capacity=2048, device=dev)
imgs = torch.stack([torch.rand(1024, 1024, generator=torch.Generator().manual_seed(s)) for s in range(a.images)]).to(dev)
pairs = pl.exhaustive_pairs(a.images, a.pairs)
out = {"workload": f"configs[3] on 1 GPU: {a.images} synthetic 1024^2 images, first {pairs.shape[0]} exhaustive pairs, 2048 keypoints", "modes": {}}
for mode, conf in (("fixed", {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}),
                   ("adaptive", {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": 1536})):
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=a.batch, max_kpts=2048, device=dev)
    pipe = pl.PairMatchingPipeline(ext, mat)
    pipe.match_all(pipe.extract_all(imgs[: a.batch]), pl.exhaustive_pairs(min(a.batch, a.images), a.batch))  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = pipe.extract_all(imgs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    cnt, mt, ms = pipe.match_all(table, pairs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    P = pairs.shape[0]
    out["modes"][mode] = {"extract_s": t1 - t0, "match_s": t2 - t1, "images_per_s": a.images / (t1 - t0), "match_pairs_per_s": P / (t2 - t1),
                          "end_to_end_pairs_per_s": P / (t2 - t0), "matches_total": int(cnt.sum().item()),
                          "conf": conf}
    del mat, pipe
print(json.dumps(out))
