#!/bin/bash
# round 6, box f: adaptive depth at one pair per call — ONE assignment pass after the layer loop (dim_tune_set key 17) and the host following the stop flags (key 18):
# config 1 through the hooks with each on / off, the bench line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -x -q -m gpu -k "lightglue or config1 or plugins or shipped" 2>&1 | tail -4
for v in "17=1 --tune 18=1" "17=1 --tune 18=0" "17=0 --tune 18=0" "17=1 --tune 18=1" "17=1 --tune 18=0"; do
timeout 300 python bench.py --workload config1 --no-cpu-baseline --tune $v > gpurun_out/r06f_config1.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06f_config1.json').read().strip().splitlines()[-1]); print('config1 tune $v', round(d['value'],1), d['hook_path'])" | cut -c1-400
done
