cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q -rfs -x --deselect tests/test_geom_verify_gpu.py::test_device_ransac_vs_the_reference_estimator_iou 2>&1 | tail -60 > gpurun_out/t2.log
bash scripts/gpu_aliked_profile.sh r03b pmc > gpurun_out/aliked_profile_r03b.log 2>&1
python scripts/gpu_aliked_bench.py 9=0 > gpurun_out/aliked_bench_r03b_nofuse.json 2>&1
python bench.py --workload config4 > gpurun_out/bench_config4_r03b.json 2> gpurun_out/bench_config4_r03b.err
python bench.py --no-cpu-baseline > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err
cat gpurun_out/t2.log | tail -30; cat gpurun_out/aliked_bench_r03b.json gpurun_out/aliked_bench_r03b_nofuse.json gpurun_out/config5_r03b.json; cat gpurun_out/bench_config4_r03b.json | cut -c1-1500; tail -2 gpurun_out/bench_config4_r03b.err; cut -c1-400 gpurun_out/bench_r03b.json
