#!/bin/bash
# same-box A/B of the LDS double-buffered activation tile in the pipelined GEMM blocks (dim_tune_set 14 = 33) + per-kernel stats of both
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
  for V in base dbuf; do
    T=""; [ $V = dbuf ] && T="--tune 14=33"
    python bench.py $T --no-cpu-baseline --no-strong-scaling --main-region-only > $O/ab_dbuf_${V}_$rep.json 2>> $O/ab_dbuf.err
    python - <<PY
import json
d = json.loads(open("$O/ab_dbuf_${V}_$rep.json").read().strip().splitlines()[-1])
print(json.dumps({"variant": "$V", "rep": $rep, "pairs_per_s": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 2), "clock_mhz": round(d["sustained_clock_mhz"])}))
PY
  done
done
bash scripts/gpu_kernel_stats.sh dbuf_base | grep -i "gemm_x6" 
bash scripts/gpu_kernel_stats.sh dbuf_on --tune 14=33 | grep -i "gemm_x6"
