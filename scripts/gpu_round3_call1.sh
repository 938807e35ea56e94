cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
python -c "import cv2; print('cv2', cv2.__version__)" > gpurun_out/probe_cv2.txt 2>&1; python -c "import h5py; print('h5py', h5py.__version__)" >> gpurun_out/probe_cv2.txt 2>&1; nproc >> gpurun_out/probe_cv2.txt
timeout 1500 python -m pytest tests/test_async_export_gpu.py tests/test_pairs_gpu.py tests/test_tile_matching_gpu.py tests/test_aliked_gpu.py tests/test_geom_verify_gpu.py "tests/test_configs_gpu.py::test_config5_aliked_full_tile_vs_oracle" -m gpu -q -rs 2>&1 | tail -40 > gpurun_out/t1.log
python scripts/gpu_end_to_end.py > gpurun_out/e2e_r03a.json 2> gpurun_out/e2e_r03a.err
bash scripts/gpu_aliked_profile.sh r03a > gpurun_out/aliked_profile_r03a.log 2>&1
cat gpurun_out/t1.log; cat gpurun_out/e2e_r03a.json; tail -3 gpurun_out/e2e_r03a.err; cat gpurun_out/probe_cv2.txt
