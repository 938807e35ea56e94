#!/bin/bash
# round 5, box o: token-confidence counts aggregated per workgroup (one atomic per 32 points instead of one per point): config 5 / 4 / 1 lines, kernel trace of config 5, LightGlue GPU tests
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 300 python bench.py --workload config5 > gpurun_out/o_config5.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/o_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['tile_pairs_per_s'], d['phases_s_max_over_ranks'] if 'phases_s_max_over_ranks' in d else '')"
timeout 300 python bench.py --workload config4 > gpurun_out/o_config4.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/o_config4.json').read().strip().splitlines()[-1]); print('config4', d['value'])"
timeout 300 python bench.py --workload config1 > gpurun_out/o_config1.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/o_config1.json').read().strip().splitlines()[-1]); print('config1', d['value'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload config5 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/o_c5_kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c5
head -6 $GRAFT_REPO_ROOT/gpurun_out/o_c5_kernel_stats.csv | cut -c1-140; grep confidence $GRAFT_REPO_ROOT/gpurun_out/o_c5_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "lightglue or config or tiled or pipeline or tile_matching" 2>&1 | tail -3
