#!/bin/bash
# Round 5, call C: config-1 real-image tests again (after the order-robust comparison), the research-build Winograd test, and a cProfile of the
# per-call plugin hooks (13 / 16 ms of wall time around 0.83 / 2.05 ms of device work in call B).
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_config1_real_gpu.py tests/test_superpoint_gpu.py -x -q -m gpu --durations=10 > $O/${TAG}_tests.log 2>&1
tail -12 $O/${TAG}_tests.log
timeout 300 python scripts/gpu_hook_profile.py > $O/${TAG}_hook_profile.txt 2>&1
head -c 6000 $O/${TAG}_hook_profile.txt
timeout 600 python bench.py --workload config1 --no-cpu-baseline > $O/${TAG}_config1.json 2> $O/${TAG}_config1.err
cp $O/parity_measured.jsonl $O/${TAG}_parity_measured.jsonl 2>/dev/null
