#!/bin/bash
# rocprofv3 kernel trace of the adaptive and the fixed-work LightGlue call of bench.measure_adaptive (6 calls each) -> gpurun_out/prof_adaptive/{ad,fx}_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adaptive -o ad -- python $R/scripts/gpu_lg_adaptive_trace.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adaptive -o fx -- python $R/scripts/gpu_lg_adaptive_trace.py fixed > /dev/null 2>&1
cd $R
python - <<PY
import csv, re
for tag in ("ad", "fx"):
    rows = list(csv.DictReader(open(f"gpurun_out/prof_adaptive/{tag}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", tag)
    for r in rows[:24]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:80]
        print(f'{n:80s} calls/6 {int(r["Calls"])/6:6.1f} per_call_us {float(r["TotalDurationNs"])/6e3:9.1f} avg_us {float(r["AverageNs"])/1e3:8.1f}')
    print("total ms per call:", tot / 6e6)
PY
