#!/bin/bash
# round 5, box n: cross-half reductions by v_permlane32_swap (product) vs ds_bpermute (variant B = -DDIM_NO_PERMLANE): headline A/B on one box, batch-1 calls
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash scripts/gpu_ab_libs.sh 2 | tee gpurun_out/n_ab.txt
timeout 200 python scripts/gpu_batch1_check.py > gpurun_out/n_batch1.json 2>/dev/null; cat gpurun_out/n_batch1.json
timeout 600 python -m pytest tests -x -q -m gpu -k "lightglue or aliked" 2>&1 | tail -3 | tee gpurun_out/n_tests.log
