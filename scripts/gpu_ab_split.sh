#!/bin/bash
# same-box A/B of the fp16 split's residual: v_fma_mix_f32 (product) vs convert + subtract (libdim_hip_nomix.so: build it first, in the build container, with
#   python -c "import importlib; importlib.import_module('deep-image-matching_amd.build').build_variant('nomix', ['-DDIM_SPLIT_NO_MIX'])")
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
  for V in mix nomix; do
    L=""; [ $V = nomix ] && L="--lib deep-image-matching_amd/lib/libdim_hip_nomix.so"
    python bench.py $L --no-cpu-baseline --no-strong-scaling --main-region-only > $O/ab_split_${V}_$rep.json 2>> $O/ab_split.err
    python - <<PY
import json
d = json.loads(open("$O/ab_split_${V}_$rep.json").read().strip().splitlines()[-1])
print(json.dumps({"variant": "$V", "rep": $rep, "pairs_per_s": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 2), "clock_mhz": round(d["sustained_clock_mhz"]), "conv1b_ms": round(d["roofline"]["avg_launch_ms"], 3)}))
PY
  done
done
