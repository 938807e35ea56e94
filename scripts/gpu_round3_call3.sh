cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_aliked_gpu.py tests/test_saturation_gpu.py "tests/test_configs_gpu.py::test_config5_aliked_full_tile_vs_oracle" "tests/test_configs_gpu.py::test_config5_batched_tile_matching_vs_the_sequential_loop_with_the_oracle_matcher" -m gpu -q -rfs 2>&1 | tail -40 > gpurun_out/t3.log
bash scripts/gpu_aliked_profile.sh r03c > gpurun_out/aliked_profile_r03c.log 2>&1
cat gpurun_out/t3.log | tail -25; cat gpurun_out/aliked_bench_r03c.json gpurun_out/config5_r03c.json; cat gpurun_out/parity_measured.jsonl
