# Final refresh after the last kernels of the round (assignment passes, similarity on the matrix cores): bench line + the four rocprofv3
# passes first, then config 4 and the end-to-end run, the whole -m gpu suite last.
TAG=${1:-r03y}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
bash scripts/gpu_collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
python bench.py --workload config4 > gpurun_out/bench_config4_$TAG.json 2> /dev/null
python scripts/gpu_end_to_end.py > gpurun_out/e2e_$TAG.json 2> gpurun_out/e2e_$TAG.err
cut -c1-300 gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_config4_$TAG.json
timeout 1200 python -m pytest tests -m gpu -q -rfs > gpurun_out/gpu_tests_$TAG.log 2>&1
tail -5 gpurun_out/gpu_tests_$TAG.log
