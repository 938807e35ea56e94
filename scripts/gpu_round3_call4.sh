cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_aliked_gpu.py "tests/test_configs_gpu.py::test_config5_aliked_full_tile_vs_oracle" -m gpu -q -rfs 2>&1 | tail -30 > gpurun_out/t4.log
python scripts/gpu_aliked_bench.py 10=8 > gpurun_out/aliked_bench_r03d_th8.json 2>/dev/null
bash scripts/gpu_aliked_profile.sh r03d > gpurun_out/aliked_profile_r03d.log 2>&1
cat gpurun_out/t4.log | tail -12; cat gpurun_out/aliked_bench_r03d.json gpurun_out/aliked_bench_r03d_th8.json gpurun_out/config5_r03d.json; cat gpurun_out/parity_measured.jsonl
