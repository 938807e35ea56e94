#!/bin/bash
# Round 5, call B: config 1 on its real inputs (GPU parity tests through the plugin hooks + the config-1 bench line) and the default bench line
# with the new hook_path sub-record (the round's starting point on this box).
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_config1_real_gpu.py -x -q -m gpu --durations=10 > $O/${TAG}_config1_tests.log 2>&1
tail -15 $O/${TAG}_config1_tests.log
timeout 600 python bench.py --workload config1 > $O/${TAG}_config1.json 2> $O/${TAG}_config1.err
tail -c 1500 $O/${TAG}_config1.json
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 600 $O/${TAG}_bench.json
cp $O/parity_measured.jsonl $O/${TAG}_parity_measured.jsonl 2>/dev/null
