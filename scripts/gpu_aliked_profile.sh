#!/bin/bash
# ALIKED evidence for profiles/ (VERDICT r2 next #5): timings, per-kernel stats and the HBM / MFMA counters at the config-5
# tile size (1500 x 1000, batch 8).  Counters in their own passes with --kernel-trace only.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R && python scripts/gpu_aliked_bench.py > $OUT/aliked_bench_$TAG.json 2> $OUT/aliked_bench_$TAG.err
python scripts/gpu_config5.py > $OUT/config5_$TAG.json 2> $OUT/config5_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_aliked_$TAG -o aliked -- python $R/scripts/gpu_aliked_one.py 8 1000 1500 3 > /dev/null 2>&1
if [ "${2:-}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_aliked_${TAG}_$C -o pmc -- python $R/scripts/gpu_aliked_one.py 8 1000 1500 1 > /dev/null 2>&1
  done
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_aliked_${TAG}_MFMA -o pmc -- python $R/scripts/gpu_aliked_one.py 8 1000 1500 1 > /dev/null 2>&1
fi
cat $OUT/aliked_bench_$TAG.json $OUT/config5_$TAG.json
