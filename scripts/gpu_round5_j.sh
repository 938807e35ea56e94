#!/bin/bash
# Round 5, call J: the fused feed-forward on 128-row blocks at the 512-register point (research build, dim_tune_set(14, 128)) vs the product kernel:
# correctness vs fp64 at production rows + kernel time (scripts/gpu_ffn_fused_check.py), then the bench A/B on the same box.
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/deep-image-matching_amd/lib/libdim_hip_research.so
for V in 32 128; do
  DIM_LIB=$L DIM_TUNE14=$V timeout 300 python scripts/gpu_ffn_fused_check.py > $O/${TAG}_ffn_check_$V.json 2> $O/${TAG}_ffn_check_$V.err
  tail -c 900 $O/${TAG}_ffn_check_$V.json; echo
done
for rep in 1 2; do
  for V in 32 128; do
    timeout 300 python bench.py --lib $L --tune 14=$V --steps 10 --warmup 2 --no-cpu-baseline --no-strong-scaling --no-hook-path --main-region-only > $O/${TAG}_bench_kc${V}_$rep.json 2>> $O/${TAG}_bench.err
    python -c "
import json
d=json.loads(open('$O/${TAG}_bench_kc${V}_$rep.json').read().strip().splitlines()[-1])
print('14=$V rep $rep', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step clock', round(d['sustained_clock_mhz']), 'guard', d['fp16x3_range_guard']['violations'])"
  done
done
