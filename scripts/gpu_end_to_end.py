"""End-to-end rate WITH geometric verification and the output containers (SURVEY §8 f2 / f3, §8(e) caveat): extraction ->
matching -> batched device RANSAC -> asynchronous features / raw_matches / matches / database.db writers, next to the
kernel-path rate of the same run.  Prints one JSON line for profiles/.

    python scripts/gpu_end_to_end.py [--images 64] [--pairs 1000]
"""
import argparse
import importlib
import json
import shutil
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
PKG = "deep-image-matching_amd"
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--pairs", type=int, default=1000)
ap.add_argument("--batch", type=int, default=50)
a = ap.parse_args()
m = lambda n: importlib.import_module(PKG + "." + n)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
weights = m("weights")
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}   # threshold 0: random weights still give match lists to verify / write
ext = m("superpoint_hip").SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=a.batch, max_hw=(1024, 1024), capacity=2048, device=dev)
mat = m("lightglue_hip").LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, gain=2.0), conf, max_pairs=a.batch, max_kpts=2048, device=dev)
imgs = torch.stack([torch.rand(1024, 1024, generator=torch.Generator().manual_seed(s)) for s in range(a.images)]).to(dev)
names = [f"img{i:04d}.jpg" for i in range(a.images)]
pairs = m("pipeline").exhaustive_pairs(a.images, a.pairs)
res = {}
for label, use_ver, use_exp in (("kernels_only", False, False), ("with_device_ransac", True, False), ("with_ransac_and_writers", True, True)):
    tmp = Path(tempfile.mkdtemp(prefix="dim_e2e_"))
    ver = m("verify").DeviceVerifier(threshold=4.0, iters=2048, device=dev) if use_ver else None
    exp = m("async_export").AsyncExporter(tmp, device=dev, image_names=names, min_inliers_per_pair=0, min_inlier_ratio_per_pair=0.0) if use_exp else None
    r = m("async_export").EndToEndRunner(ext, mat, ver, exp).run(names, imgs, pairs)
    if label == "kernels_only":   # first run also warms up: repeat
        r = m("async_export").EndToEndRunner(ext, mat, None, None).run(names, imgs, pairs)
    res[label] = r
    shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps({"workload": f"{a.images} synthetic 1024^2 images, {pairs.shape[0]} pairs, 2048 keypoints, fixed-work LightGlue, threshold 0 (all mutual matches written)",
                  "runs": res}))
