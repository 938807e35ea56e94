cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
timeout 1500 python -m pytest tests/test_lightglue_gpu.py tests/test_aliked_gpu.py tests/test_saturation_gpu.py "tests/test_configs_gpu.py::test_config5_aliked_full_tile_vs_oracle" "tests/test_configs_gpu.py::test_config4_exhaustive_pairs_through_the_pipeline_vs_oracle" -m gpu -q -rfs 2>&1 | tail -40 > gpurun_out/t5.log
python bench.py --no-cpu-baseline --main-region-only --tune 11=0 > gpurun_out/bench_r03e_noln.json 2> /dev/null
python bench.py --no-cpu-baseline --main-region-only > gpurun_out/bench_r03e_ln.json 2> /dev/null
python bench.py --no-cpu-baseline --main-region-only --tune 11=0 > gpurun_out/bench_r03e_noln2.json 2> /dev/null
python bench.py --no-cpu-baseline --main-region-only > gpurun_out/bench_r03e_ln2.json 2> /dev/null
python scripts/gpu_aliked_bench.py > gpurun_out/aliked_bench_r03e.json 2>/dev/null
python scripts/gpu_config5.py > gpurun_out/config5_r03e.json 2>/dev/null
cat gpurun_out/t5.log | tail -25
for f in noln ln noln2 ln2; do python -c "import json;d=json.load(open('gpurun_out/bench_r03e_$f.json'));print('$f', round(d['value'],1), round(d['ms_per_step'],2), round(d['sustained_clock_mhz']))"; done
cat gpurun_out/aliked_bench_r03e.json gpurun_out/config5_r03e.json
