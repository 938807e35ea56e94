cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_aliked_gpu.py "tests/test_configs_gpu.py::test_config5_aliked_full_tile_vs_oracle" -m gpu -q 2>&1 | tail -3
bash scripts/gpu_aliked_profile.sh r03i > gpurun_out/aliked_profile_r03i.log 2>&1
cat gpurun_out/aliked_bench_r03i.json
