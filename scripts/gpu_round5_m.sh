#!/bin/bash
# round 5, box m: batch-1 attention with two key tiles in flight vs one (A/B inside one process), kernel trace of the batch-1 calls
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 python scripts/gpu_batch1_check.py > gpurun_out/batch1_m.json 2> gpurun_out/batch1_m.err; tail -2 gpurun_out/batch1_m.err; cat gpurun_out/batch1_m.json
timeout 600 python -m pytest tests -x -q -m gpu -k "lightglue or config1" 2>&1 | tail -4 > gpurun_out/m_tests.log; cat gpurun_out/m_tests.log
