#!/bin/bash
# round 5, box k: the streaming small-problem GEMM (batch-1 LightGlue through the hooks) vs the staged loop; prefetch depth 4 vs 8
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 python scripts/gpu_small_gemm_check.py > gpurun_out/small_gemm_d4.json 2> gpurun_out/small_gemm_d4.err
DIM_LIB=deep-image-matching_amd/lib/libdim_hip_stream8.so timeout 300 python scripts/gpu_small_gemm_check.py > gpurun_out/small_gemm_d8.json 2> gpurun_out/small_gemm_d8.err
tail -3 gpurun_out/small_gemm_d4.err; cat gpurun_out/small_gemm_d4.json; cat gpurun_out/small_gemm_d8.json
