#!/bin/bash
# Round-6 closing run (ONE gpurun call): the bench line + its four rocprofv3 passes (-> scripts/make_profiles.py r06 100), the co-execution counters,
# config 1 / config 4 / config 5 lines, the end-to-end run, the hook-path kernel trace, the whole -m gpu suite with durations.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round6_final.sh r06'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
mkdir -p $O; rm -f $O/parity_measured.jsonl
bash scripts/gpu_collect_profiles.sh $TAG > $O/collect_$TAG.log 2>&1
cut -c1-300 $O/bench_$TAG.json
timeout 300 python bench.py --workload config1 > $O/config1_$TAG.json 2> $O/config1_$TAG.err; cut -c1-200 $O/config1_$TAG.json
timeout 300 python bench.py --workload config4 > $O/config4_$TAG.json 2> $O/config4_$TAG.err; cut -c1-200 $O/config4_$TAG.json
timeout 300 python bench.py --workload config5 > $O/config5_$TAG.json 2> $O/config5_$TAG.err; cut -c1-200 $O/config5_$TAG.json
timeout 300 python scripts/gpu_end_to_end.py > $O/end_to_end_$TAG.json 2> $O/end_to_end_$TAG.err; tail -c 600 $O/end_to_end_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${TAG}_coexec -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c5 -o c5 -- python $R/bench.py --workload config5 > /dev/null 2>&1
f=$(find $O/prof_${TAG}_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_config5_kernel_stats.csv; rm -rf $O/prof_${TAG}_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_b1 -o b1 -- python $R/scripts/gpu_b1_bench.py > $O/b1_$TAG.json 2>/dev/null
f=$(find $O/prof_${TAG}_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_b1_kernel_stats.csv; rm -rf $O/prof_${TAG}_b1
cd $R
timeout 300 python scripts/gpu_b1_steps.py > $O/${TAG}_b1_steps.json 2>/dev/null; cut -c1-600 $O/${TAG}_b1_steps.json
timeout 300 python scripts/gpu_b1_adaptive.py 2>/dev/null | tail -1 > $O/${TAG}_b1_adaptive.json; cut -c1-400 $O/${TAG}_b1_adaptive.json
bash scripts/gpu_adaptive_trace.sh > $O/adaptive_trace_$TAG.log 2>&1; tail -3 $O/adaptive_trace_$TAG.log
SECONDS=0
timeout 1200 python -m pytest tests -m gpu -q -rfs --durations=25 > $O/gpu_tests_$TAG.log 2>&1
echo "suite wall seconds: $SECONDS" >> $O/gpu_tests_$TAG.log
tail -8 $O/gpu_tests_$TAG.log
cp $O/parity_measured.jsonl $O/${TAG}_parity_measured_suite.jsonl 2>/dev/null
