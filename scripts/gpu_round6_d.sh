#!/bin/bash
# round 6, box d: the one-pair 32 x 128 GEMM blocks with chunk-ahead requests
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 python scripts/gpu_b1_steps.py > gpurun_out/r06_b1_steps.json 2> gpurun_out/r06_b1_steps.err; cat gpurun_out/r06_b1_steps.json; tail -2 gpurun_out/r06_b1_steps.err
timeout 300 python scripts/gpu_small_gemm_variants.py > gpurun_out/r06_small_gemm_variants2.json 2>/dev/null; cat gpurun_out/r06_small_gemm_variants2.json
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06b1 -o b1 -- python $R/scripts/gpu_b1_bench.py > $R/gpurun_out/r06_b1_bench.json 2>/dev/null; cd $R
cat gpurun_out/r06_b1_bench.json
head -12 gpurun_out/prof_r06b1/b1_kernel_stats.csv | cut -c1-150
timeout 600 python -m pytest tests -x -q -m gpu -k "lightglue" 2>&1 | tail -3
