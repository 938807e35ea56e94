#!/bin/bash
# SQ counters of the conv1b variants (dim_tune_set key 15), one rocprofv3 --pmc pass per counter group and variant.
TAG=${1:-r04f}; VARS=${2:-"0 1 3"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/${TAG}_sq_counters.txt
for V in $VARS; do
  i=0
  for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/pmc_${TAG}_v${V}_g$i -o pmc -- python $R/bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=$V --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
  done
done
cd $R
python - <<'PY'
import glob, os, csv, collections, re
O = "gpurun_out"; TAG = os.environ.get("TAG", "r04f")
for d in sorted(glob.glob(f"{O}/pmc_{TAG}_v*_g*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(fs[0])):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:50]
        if "conv3x3" not in n: continue
        n += " g" + str(int(r["Grid_Size"]) // 256)
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for n in sorted(agg, key=lambda n: -dur[n])[:1]:
        c = cnt[n]; print(os.path.basename(d), n, f"calls {c} avg_us {dur[n]/c/1e3:.1f} " + " ".join(f"{k}={v/c:.4g}" for k, v in agg[n].items()))
PY
