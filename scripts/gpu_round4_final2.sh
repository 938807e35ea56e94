#!/bin/bash
# Round-4 closing run after the fused feed-forward's K loop changed: the bench line + its four rocprofv3 passes (-> make_profiles.py), the phase
# timers of the old (14 = 36) and the new loop, their kernel times on this box, the whole -m gpu suite.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round4_final2.sh r04'
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
bash scripts/gpu_collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
cut -c1-300 gpurun_out/bench_$TAG.json
( python scripts/gpu_ffn_phases.py 36 2>&1 | tail -1; python scripts/gpu_ffn_phases.py 2>&1 | tail -1 ) > gpurun_out/ffn_phases_$TAG.jsonl
bash scripts/gpu_kernel_stats.sh ${TAG}_step --tune 14=36 | grep -i "ffn_fused"
grep -i "ffn_fused" gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-120
timeout 1100 python -m pytest tests -m gpu -q -rfs > gpurun_out/gpu_tests_$TAG.log 2>&1
tail -6 gpurun_out/gpu_tests_$TAG.log
