# Round-3 evidence in ONE gpurun call: the whole -m gpu suite, the bench line + its rocprofv3 passes, ALIKED timings / profile / counters,
# config 4 and config 5 lines, the end-to-end run with writers.  TAG = suffix of the gpurun_out files (profiles/ are built from them).
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q -rfs > gpurun_out/gpu_tests_$TAG.log 2>&1
tail -5 gpurun_out/gpu_tests_$TAG.log
python scripts/gpu_aliked_bench.py 10=17 > gpurun_out/aliked_bench_${TAG}_stream.json 2>/dev/null
bash scripts/gpu_aliked_profile.sh $TAG pmc > gpurun_out/aliked_profile_$TAG.log 2>&1
python scripts/gpu_end_to_end.py > gpurun_out/e2e_$TAG.json 2> gpurun_out/e2e_$TAG.err
python bench.py --workload config4 > gpurun_out/bench_config4_$TAG.json 2> /dev/null
python scripts/gpu_ffn_fused_check.py > gpurun_out/ffn_fused_$TAG.json 2> /dev/null
python scripts/gpu_gemm_probe.py 0 1 8 4 16 32 0 > gpurun_out/gemm_probe_$TAG.json 2> /dev/null
bash scripts/gpu_collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
cat gpurun_out/aliked_bench_$TAG.json gpurun_out/aliked_bench_${TAG}_stream.json gpurun_out/config5_$TAG.json
cut -c1-300 gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_config4_$TAG.json
python -c "import json;d=json.load(open('gpurun_out/e2e_$TAG.json'))['runs'];print({k:(round(v['kernel_path_pairs_per_s']),round(v['end_to_end_pairs_per_s'])) for k,v in d.items()})"
