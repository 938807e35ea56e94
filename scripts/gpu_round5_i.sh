#!/bin/bash
# Round 5, call I: 256-row GEMM blocks at the 512-register point (research build, dim_tune_set(14, 256)) vs the product blocks, same box:
# bench (twice each, alternating) + rocprofv3 kernel stats of both.
TAG=${1:-r05i}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/deep-image-matching_amd/lib/libdim_hip_research.so
for rep in 1 2; do
  for V in 32 256; do
    timeout 300 python bench.py --lib $L --tune 14=$V --steps 10 --warmup 2 --no-cpu-baseline --no-strong-scaling --no-hook-path --main-region-only > $O/${TAG}_bench_kc${V}_$rep.json 2>> $O/${TAG}_bench.err
    python -c "
import json
d=json.loads(open('$O/${TAG}_bench_kc${V}_$rep.json').read().strip().splitlines()[-1])
print('14=$V rep $rep', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step clock', round(d['sustained_clock_mhz']), 'guard', d['fp16x3_range_guard']['violations'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for V in 32 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$V -o bench -- python $R/bench.py --lib $L --tune 14=$V --steps 3 --warmup 1 --no-cpu-baseline --no-strong-scaling --no-hook-path --main-region-only > /dev/null 2>&1
  f=$(find $O/prof_${TAG}_$V -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_kc$V.csv
  rm -rf $O/prof_${TAG}_$V
  grep -i "gemm_x6" $O/${TAG}_kernel_stats_kc$V.csv | cut -c1-200 | head -8
done
