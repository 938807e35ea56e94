#!/bin/bash
# same-box A/B of one dim_tune_set variant against the default: bash scripts/gpu_ab_tune.sh TAG KEY=VALUE [kernel-name-filter]
TAG=$1; TUNE=$2; FILT=${3:-gemm_x6}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
  for V in base $TAG; do
    T=""; [ $V != base ] && T="--tune $TUNE"
    python bench.py $T --no-cpu-baseline --no-strong-scaling --main-region-only > $O/ab_${TAG}_${V}_$rep.json 2>> $O/ab_$TAG.err
    python - <<PY
import json
d = json.loads(open("$O/ab_${TAG}_${V}_$rep.json").read().strip().splitlines()[-1])
print(json.dumps({"variant": "$V", "rep": $rep, "pairs_per_s": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 2), "clock_mhz": round(d["sustained_clock_mhz"])}))
PY
  done
done
bash scripts/gpu_kernel_stats.sh ${TAG}_base | grep -i "$FILT"
bash scripts/gpu_kernel_stats.sh ${TAG}_on --tune $TUNE | grep -i "$FILT"
