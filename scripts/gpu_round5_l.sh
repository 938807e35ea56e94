#!/bin/bash
# round 5, box l: batch-1 calls after the top-k / column-pass changes; key-split sweep of the batch-1 attention; GPU suite on the new kernels
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 python scripts/gpu_batch1_check.py > gpurun_out/batch1.json 2> gpurun_out/batch1.err; tail -2 gpurun_out/batch1.err; cat gpurun_out/batch1.json
timeout 900 python -m pytest tests -x -q -m gpu -k "superpoint or lightglue or config1 or configs" 2>&1 | tail -6 > gpurun_out/l_tests.log; cat gpurun_out/l_tests.log
timeout 300 python bench.py --no-strong-scaling --no-cpu-baseline > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/l_bench.json").read().strip().splitlines()[-1])
print(d["value"], d.get("hook_path"))
PY
