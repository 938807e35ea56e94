#!/bin/bash
# A/B sweep of the Winograd conv1b variants (dim_tune_set key 15) against the direct kernel on ONE box: bench.py main region only.
# usage: bash scripts/gpu_wino_sweep.sh TAG "0 1 17 ..."
TAG=${1:-sweep}; shift
VARS=${1:-"0 1 17 257 273 0"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for V in $VARS; do
  python bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=$V --no-cpu-baseline --no-strong-scaling --main-region-only --steps 10 --warmup 2 > $O/${TAG}_v$V.json 2>> $O/${TAG}.err
  python - <<PY
import json
d = json.loads(open("$O/${TAG}_v$V.json").read().strip().splitlines()[-1])
print("tune15=$V", round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 2), "ms/step clock", round(d["sustained_clock_mhz"]), "conv1b ms", round(d["roofline"]["avg_launch_ms"], 3), "guard", d["fp16x3_range_guard"]["violations"])
PY
done
