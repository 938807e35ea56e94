#!/bin/bash
# Round-4 evidence in ONE gpurun call: bench line (+ strong_scaling sub-record) and its four rocprofv3 passes, config 4 / config 5 lines,
# the end-to-end run, the MFMA / VALU overlap probe, the whole -m gpu suite.  TAG = suffix of the gpurun_out files.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round4_final.sh r04'   then   python scripts/make_profiles.py r04 100
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
bash scripts/gpu_collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
python bench.py --workload config4 > gpurun_out/bench_config4_$TAG.json 2> /dev/null
python bench.py --workload config5 > gpurun_out/bench_config5_$TAG.json 2> gpurun_out/bench_config5_$TAG.err
python scripts/gpu_end_to_end.py > gpurun_out/e2e_$TAG.json 2> gpurun_out/e2e_$TAG.err
[ -x scripts/probe/ovl_probe ] && ./scripts/probe/ovl_probe > gpurun_out/ovl_probe_$TAG.jsonl 2>&1
cut -c1-400 gpurun_out/bench_$TAG.json; cut -c1-300 gpurun_out/bench_config4_$TAG.json; cut -c1-300 gpurun_out/bench_config5_$TAG.json
timeout 1500 python -m pytest tests -m gpu -q -rfs > gpurun_out/gpu_tests_$TAG.log 2>&1
tail -6 gpurun_out/gpu_tests_$TAG.log
