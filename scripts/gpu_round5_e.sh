#!/bin/bash
# Round 5, call E: config 5 with a preselector that votes (block-noise first band): GPU test vs the oracle chain + bench line; hook profile; default bench.
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_tiled_pipeline_gpu.py tests/test_config1_real_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu --durations=8 > $O/${TAG}_tests.log 2>&1
tail -14 $O/${TAG}_tests.log
timeout 600 python bench.py --workload config5 > $O/${TAG}_config5.json 2> $O/${TAG}_config5.err
tail -c 1500 $O/${TAG}_config5.json; tail -3 $O/${TAG}_config5.err
timeout 300 python scripts/gpu_hook_profile.py > $O/${TAG}_hook_profile.txt 2>&1
grep "ms per call" $O/${TAG}_hook_profile.txt
timeout 600 python bench.py --workload config1 > $O/${TAG}_config1.json 2> $O/${TAG}_config1.err
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 400 $O/${TAG}_bench.json
cp $O/parity_measured.jsonl $O/${TAG}_parity_measured.jsonl 2>/dev/null
