#!/bin/bash
# Round 4, GPU call C: config 5 through bench.py (PRESELECTION and GRID), end-to-end with RANSAC on its own stream, new parity tests
TAG=${1:-r04c}; O=gpurun_out; mkdir -p $O
python bench.py --workload config5 --images 4 > $O/${TAG}_config5_presel.json 2> $O/${TAG}_config5.err; tail -c 300 $O/${TAG}_config5.err
python bench.py --workload config5 --images 4 --tile-selection GRID > $O/${TAG}_config5_grid.json 2>> $O/${TAG}_config5.err
python scripts/gpu_end_to_end.py > $O/${TAG}_end_to_end.json 2> $O/${TAG}_e2e.err; tail -c 300 $O/${TAG}_e2e.err
timeout 1200 python -m pytest tests/test_geom_verify_gpu.py tests/test_configs_gpu.py tests/test_lightglue_gpu.py -m gpu -q -x -k "ground_truth or true_correspondences or flip_rate" > $O/${TAG}_new_tests.log 2>&1; tail -5 $O/${TAG}_new_tests.log
python - <<'PY'
import json, os
T = os.environ.get("TAG", "r04c")
for f in ("config5_presel", "config5_grid"):
    try:
        d = json.loads(open(f"gpurun_out/{T}_{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 3), "pairs/s", d["phases_s_max_over_ranks"], "kpts", d["keypoints_per_image_mean"], "matches/pair", d["matches_per_pair_mean"], d["pairs_with_matches"], "roofline", round(d["roofline"]["achieved"]), "GB/s", round(d["roofline"]["avg_launch_ms"], 3), "ms")
    except Exception as e: print(f, "ERR", e)
try:
    d = json.load(open(f"gpurun_out/{T}_end_to_end.json"))["runs"]
    print({k: (round(v.get("kernel_path_pairs_per_s", 0)), round(v.get("end_to_end_pairs_per_s", 0))) for k, v in d.items() if "pairs" in v and "kernel_path_pairs_per_s" in v}, d.get("device_ransac_alone"))
except Exception as e: print("e2e ERR", e)
PY
