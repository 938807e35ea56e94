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
# real work for the verifier and the writers: every image is a crop of ONE canvas at a multiple-of-8 offset and the
# LightGlue weights are the matching-capable synthetic set, so a pair carries hundreds of true (pure-translation) correspondences
# above the reference's default threshold 0.1
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
ext = m("superpoint_hip").SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=a.batch, max_hw=(1024, 1024), capacity=2048, device=dev)
imgs_cpu, offsets = m("workloads").shifted_crops(a.images, 1024, 1024, max_shift=256, seed=7)
imgs = imgs_cpu.to(dev)
center = m("workloads").descriptor_mean(ext, imgs)
mat = m("lightglue_hip").LightGlueHIP(weights.synthetic_lightglue_matching_state_dict(0, 256, center=center), conf, max_pairs=a.batch, max_kpts=2048, device=dev)
names = [f"img{i:04d}.jpg" for i in range(a.images)]
pairs = m("pipeline").exhaustive_pairs(a.images, a.pairs)
res = {}
for label, use_ver, use_exp, ovl in (("kernels_only", False, False, True), ("with_device_ransac_same_stream", True, False, False), ("with_device_ransac", True, False, True),
                                    ("with_ransac_and_writers", True, True, True)):
    tmp = Path(tempfile.mkdtemp(prefix="dim_e2e_"))
    ver = m("verify").DeviceVerifier(threshold=4.0, iters=2048, device=dev) if use_ver else None
    exp = m("async_export").AsyncExporter(tmp, device=dev, image_names=names, min_inliers_per_pair=0, min_inlier_ratio_per_pair=0.0) if use_exp else None
    r = m("async_export").EndToEndRunner(ext, mat, ver, exp, overlap_verification=ovl).run(names, imgs, pairs)
    if label == "kernels_only":   # first run also warms up: repeat
        r = m("async_export").EndToEndRunner(ext, mat, None, None).run(names, imgs, pairs)
    r["bytes_written"] = sum(f.stat().st_size for f in tmp.rglob("*") if f.is_file())
    res[label] = r
    shutil.rmtree(tmp, ignore_errors=True)
# the device RANSAC alone, on one matched batch: ms per pair at this match count
table = ext.extract_batch(imgs[: min(a.images, a.batch)].contiguous())
pp = pairs[: a.batch].to(dev, torch.int32).contiguous()
pp = pp[(pp < min(a.images, a.batch)).all(1)].contiguous()
size = torch.full((a.images, 2), 1024.0, device=dev)
o = mat.match_batch(table[0], table[2], table[3], size, pair_idx=pp)
ver = m("verify").DeviceVerifier(threshold=4.0, iters=2048, device=dev)
ver.verify_batch(table[0], o["matches"], o["n_matches"], pair_idx=pp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    v = ver.verify_batch(table[0], o["matches"], o["n_matches"], pair_idx=pp)
e1.record()
torch.cuda.synchronize()
nm, ni = o["n_matches"].cpu(), v["n_inliers"].cpu()
# how many of the matches are the TRUE correspondences (pure translation between the two crops)
kp_cpu, mt_cpu = table[0].cpu(), o["matches"].cpu()
frac = [m("workloads").true_match_fraction(kp_cpu[i], kp_cpu[j], mt_cpu[q, : int(nm[q])], offsets[i], offsets[j]) for q, (i, j) in enumerate(pp.cpu().tolist())]
res["device_ransac_alone"] = {"pairs": int(pp.shape[0]), "ms_per_pair": e0.elapsed_time(e1) / 5 / max(1, int(pp.shape[0])), "iters": 2048,
                              "matches_per_pair_mean": float(nm.float().mean()), "matches_per_pair_min": int(nm.min()), "inliers_per_pair_mean": float(ni.float().mean()),
                              "true_correspondence_fraction_mean": sum(frac) / max(1, len(frac))}
print(json.dumps({"workload": f"{a.images} crops (1024^2, multiple-of-8 offsets <= 256 px) of one synthetic canvas, {pairs.shape[0]} pairs, 2048 keypoints, fixed-work LightGlue "
                              "with matching-capable synthetic weights, threshold 0.1",
                  "runs": res}))
