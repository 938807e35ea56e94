#!/bin/bash
# SQ_VALU_MFMA_COEXEC_CYCLES (cycles in which the vector ALU and the matrix pipe of a SIMD work at the same time) for the bench kernels and for
# the overlap probe: the direct check of DESIGN.md section 5's "MFMA and VALU do not overlap on a SIMD".
TAG=${1:-r04co}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${TAG}_bench -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${TAG}_bench2 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${TAG}_probe -o pmc -- $R/scripts/probe/ovl_probe > /dev/null 2>&1
cd $R
python - <<'PY'
import glob, os, csv, collections, re
O = "gpurun_out"; TAG = os.environ.get("TAG", "r04co")
for d in sorted(glob.glob(f"{O}/pmc_{TAG}_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(fs[0])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace("void ", "")[:56]
        n += " g" + str(int(r["Grid_Size"]) // 256)
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for n in sorted(agg, key=lambda n: -dur[n])[:14]:
        c = cnt[n]; a = agg[n]
        line = f"{os.path.basename(d)} {n}: calls {c} avg_us {dur[n]/c/1e3:.1f} "
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in a and a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            cyc = a["GRBM_GUI_ACTIVE"] / c / 8
            line += f"mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES']/c/(1024*cyc):.3f} valu_active {4*a['SQ_ACTIVE_INST_VALU']/c/(1024*cyc):.3f} coexec/mfma_busy {a['SQ_VALU_MFMA_COEXEC_CYCLES']/a['SQ_VALU_MFMA_BUSY_CYCLES']:.3f} coexec_frac_of_time {a['SQ_VALU_MFMA_COEXEC_CYCLES']/c/(1024*cyc):.3f}"
        else:
            line += " ".join(f"{k}={v/c:.4g}" for k, v in sorted(a.items()))
        print(line)
PY
