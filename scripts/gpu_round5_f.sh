#!/bin/bash
# Round 5, call F: the whole -m gpu suite with per-test durations (keep it under 600 s) + the HIP side of the round-5 flip-rate
# study (threshold 0.1, adaptive depth / width; the oracle side ran in the build container).
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_measured.jsonl
timeout 300 python scripts/study/lg_flip_rate.py gpu2 200 > $O/${TAG}_flip2_gpu.log 2>&1; tail -2 $O/${TAG}_flip2_gpu.log
SECONDS=0
timeout 1500 python -m pytest tests/ -q -m gpu --durations=45 > $O/${TAG}_gpu_tests.log 2>&1
echo "suite wall seconds: $SECONDS" >> $O/${TAG}_gpu_tests.log
tail -60 $O/${TAG}_gpu_tests.log
cp $O/parity_measured.jsonl $O/${TAG}_parity_measured.jsonl 2>/dev/null
