#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench main region -> gpurun_out/prof_$TAG/bench_kernel_stats.csv + a printed top list
TAG=${1:-r04k}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --main-region-only "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv, re
rows = list(csv.DictReader(open("gpurun_out/prof_$TAG/bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:70]
    print(f'{n:70s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):6.2f}')
print("total ms per step (3 steps + ...):", tot / 1e6)
PY
