#!/bin/bash
# Round 5, call H: the filler probe in the POWER-LIMITED regime (toggling operands, 0.3 s per line): do fillers that hide in cycles also hide in
# wall time?  + the oracle-heavy GPU tests again with the oracle on 16 threads (durations).
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done) > $O/${TAG}_smi.txt 2>&1 &
timeout 120 $R/scripts/probe/filler_probe 300000 1 > $O/${TAG}_filler_probe_power.jsonl 2> $O/${TAG}_filler_probe_power.err
wait
cat $O/${TAG}_filler_probe_power.jsonl
cd $R
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_pairs_gpu.py tests/test_lightglue_gpu.py -q -m gpu --durations=12 > $O/${TAG}_tests.log 2>&1
tail -18 $O/${TAG}_tests.log
