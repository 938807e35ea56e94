#!/bin/bash
# Round 5, call A: (1) MFMA-gap filler probe at ONE wave per SIMD (VERDICT r4 next #2a), raw + co-execution counters;
# (2) the batch-1 hook-shaped calls: wall time per call vs the sum of their kernel times (rocprofv3 kernel trace) — how much of the hook
# path is launch gaps.
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/scripts/probe/filler_probe 4000 > $O/${TAG}_filler_probe.jsonl 2> $O/${TAG}_filler_probe.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${TAG}_filler -o pmc -- $R/scripts/probe/filler_probe 1000 > /dev/null 2>&1
timeout 300 python $R/scripts/gpu_b1_bench.py > $O/${TAG}_b1.json 2> $O/${TAG}_b1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_b1 -o b1 -- python $R/scripts/gpu_b1_bench.py > /dev/null 2>&1
cd $R
python - <<PY
import glob, csv, collections, re, json
O = "gpurun_out"; TAG = "$TAG"
fs = glob.glob(f"{O}/pmc_{TAG}_filler/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set(); order = []
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name'].split('(')[0].replace("void ", "")
        if n not in agg: order.append(n)
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    with open(f"{O}/{TAG}_filler_pmc.txt", "w") as f:
        for n in order:
            a = agg[n]
            if a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
                f.write(f"{n}: mfma_busy_cycles {a['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} coexec_cycles {a['SQ_VALU_MFMA_COEXEC_CYCLES']:.4g} coexec/mfma_busy {a['SQ_VALU_MFMA_COEXEC_CYCLES']/a['SQ_VALU_MFMA_BUSY_CYCLES']:.3f} valu_inst {a['SQ_ACTIVE_INST_VALU']:.4g} wave_cycles {a['SQ_WAVE_CYCLES']:.4g} gui_active {a['GRBM_GUI_ACTIVE']:.4g}\n")
            else:
                f.write(f"{n}: " + " ".join(f"{k}={v:.4g}" for k, v in sorted(a.items())) + "\n")
fs = glob.glob(f"{O}/prof_{TAG}_b1/**/*kernel_stats.csv", recursive=True)
if fs:
    import shutil; shutil.copy(fs[0], f"{O}/{TAG}_b1_kernel_stats.csv")
fs = glob.glob(f"{O}/prof_{TAG}_b1/**/*kernel_trace.csv", recursive=True)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # gaps between consecutive kernels, and busy time, over the traced run
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
    json.dump({"kernels": len(rows), "busy_ms": busy / 1e6, "span_ms": span / 1e6}, open(f"{O}/{TAG}_b1_trace_summary.json", "w"))
PY
rm -rf $O/pmc_${TAG}_filler $O/prof_${TAG}_b1
