#!/bin/bash
# round 5, box q: the LightGlue hook with one upload and one download per call; pruning without identity copies; config 1 / 5 lines; GPU tests of the hooks
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests -x -q -m gpu -k "config1 or plugins or hooks or reference_base or loader or lightglue" 2>&1 | tail -4
timeout 300 python bench.py --workload config1 > gpurun_out/q_config1.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/q_config1.json').read().strip().splitlines()[-1]); print('config1', d['value'], {k: v for k, v in d.items() if 'hook' in k or 'batched' in k})" | cut -c1-600
timeout 300 python bench.py --workload config5 > gpurun_out/q_config5.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/q_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['tile_pairs_per_s'])"
timeout 300 python bench.py --no-strong-scaling --no-cpu-baseline --no-live-traffic > gpurun_out/q_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/q_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['hook_path'])" | cut -c1-500
