#!/bin/bash
# Round 4, GPU call A (one gpurun call): headline bench (with the strong_scaling sub-record), the 64-wide-K-chunk GEMM prototype
# (dim_tune_set(14, 64), unmeasured in round 3) A/B on the same box, the 200-pair match-list flip study (HIP side), the end-to-end run
# with real correspondences, and the new trained-checkpoint / true-correspondence parity tests.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round4_a.sh r04a'
TAG=${1:-r04a}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -c 600 $O/${TAG}_bench.err
python bench.py --lib deep-image-matching_amd/lib/libdim_hip_research.so --tune 14=64 --no-cpu-baseline --no-strong-scaling > $O/${TAG}_bench_kc64.json 2>> $O/${TAG}_bench.err
python bench.py --no-cpu-baseline --no-strong-scaling > $O/${TAG}_bench_again.json 2>> $O/${TAG}_bench.err
python scripts/study/lg_flip_rate.py gpu 200 > $O/${TAG}_flip_gpu.log 2>&1; tail -2 $O/${TAG}_flip_gpu.log
python scripts/gpu_end_to_end.py > $O/${TAG}_end_to_end.json 2> $O/${TAG}_e2e.err; tail -c 400 $O/${TAG}_e2e.err
timeout 900 python -m pytest tests/test_aliked_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "trained or true_correspondences" > $O/${TAG}_new_tests.log 2>&1; tail -5 $O/${TAG}_new_tests.log
TAG=$TAG python - <<'PY'
import json,sys,os
for f in ("bench","bench_kc64","bench_again"):
    try:
        d=json.loads(open(f"gpurun_out/%s_%s.json" % (os.environ.get("TAG","r04a"), f)).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), "pairs/s", round(d["ms_per_step"],2), "ms/step clock", round(d["sustained_clock_mhz"]), "conv1b", round(d["roofline"]["avg_launch_ms"],2), "strong", (d.get("strong_scaling") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
