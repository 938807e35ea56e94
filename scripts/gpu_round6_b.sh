#!/bin/bash
# round 6, box b: the one-pair (plugin-hook) LightGlue path after the 32 x 128 GEMM block, the in-kernel attention merge, K | V images and LayerNorm + GELU fused
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -x -q -m gpu -k "lightglue or config1 or plugins" 2>&1 | tail -4
timeout 300 python scripts/gpu_b1_steps.py > gpurun_out/r06_b1_steps.json 2> gpurun_out/r06_b1_steps.err; cat gpurun_out/r06_b1_steps.json; tail -2 gpurun_out/r06_b1_steps.err
timeout 300 python scripts/gpu_small_gemm_variants.py > gpurun_out/r06_small_gemm_variants2.json 2>/dev/null; cat gpurun_out/r06_small_gemm_variants2.json
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06b1 -o b1 -- python $R/scripts/gpu_b1_bench.py > $R/gpurun_out/r06_b1_bench.json 2>/dev/null; cd $R
cat gpurun_out/r06_b1_bench.json
head -24 gpurun_out/prof_r06b1/b1_kernel_stats.csv | cut -c1-150
