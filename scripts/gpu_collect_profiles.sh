#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line + the four rocprofv3 passes that profiles/ is built from
# (scripts/make_profiles.py <tag> turns gpurun_out/ into the committed summaries).  Counters are collected in
# their own passes with --kernel-trace only, as the pool requires.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R && python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 600 $OUT/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_${TAG}_MFMA -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
find $OUT -name "*.csv" | head -20
