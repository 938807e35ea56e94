#!/bin/bash
# round 6, box e: bench lines after the one-pair changes (headline + hook_path, config 1 through the hooks); the pair hook uploads the arrays as they are
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -x -q -m gpu -k "config1 or plugins or hooks or reference_base or loader or lightglue or shipped" 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r06e_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['hook_path'])" | cut -c1-900
timeout 300 python bench.py --workload config1 > gpurun_out/r06e_config1.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06e_config1.json').read().strip().splitlines()[-1]); print('config1', d['value'], {k: v for k, v in d.items() if 'hook' in k or 'batched' in k})" | cut -c1-900
