cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest "tests/test_lightglue_gpu.py::test_lightglue_gpu_full_size_kv_images_from_the_projection_gemm" -m gpu -q -x 2>&1 | grep -v "^    \|^$" | tail -60 > gpurun_out/t6.log
python scripts/gpu_config5.py > gpurun_out/config5_r03f.json 2> gpurun_out/config5_r03f.err
cat gpurun_out/t6.log; tail -5 gpurun_out/config5_r03f.err; cat gpurun_out/config5_r03f.json
