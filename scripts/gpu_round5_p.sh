#!/bin/bash
# round 5, box p: kernel traces of the config-4 and config-1 lines (looking for kernels that cost more than they should off the headline path)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in config4 config1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o p -- python $R/bench.py --workload $W > $R/gpurun_out/p_$W.json 2>/dev/null
  f=$(find $R/gpurun_out/prof_$W -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/p_${W}_kernel_stats.csv; rm -rf $R/gpurun_out/prof_$W
  python -c "
import json; d=json.loads(open('$R/gpurun_out/p_$W.json').read().strip().splitlines()[-1]); print('$W', d['value'])"
done
