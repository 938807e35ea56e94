R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R; python bench.py --tune 2=1 --no-cpu-baseline --no-strong-scaling --main-region-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tune 2=1 (8-row tiles, 3 wg/CU):', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step', round(d['sustained_clock_mhz']))"
python bench.py --no-cpu-baseline --no-strong-scaling --main-region-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step', round(d['sustained_clock_mhz']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_r04mr2 -o pmc -- python $R/bench.py --tune 2=1 --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
cd $R
python - <<'PY'
import glob, csv, collections, re
fs = glob.glob("gpurun_out/pmc_r04mr2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(fs[0])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace("void ", "").replace(", ", ",")[:52] + " g" + str(int(r["Grid_Size"]) // 256)
    if "conv3x3" not in n: continue
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n in sorted(agg, key=lambda n: -dur[n]):
    c = cnt[n]; a = agg[n]; cyc = a["GRBM_GUI_ACTIVE"] / c / 8
    mb = a['SQ_VALU_MFMA_BUSY_CYCLES'] / c / (1024 * cyc); va = 4 * a['SQ_ACTIVE_INST_VALU'] / c / (1024 * cyc); co = a['SQ_VALU_MFMA_COEXEC_CYCLES'] / c / (1024 * cyc)
    print("%-60s calls %d avg_us %9.1f mfma_busy %.3f valu_active %.3f coexec %.3f waves/SIMD %.2f" % (n, c, dur[n] / c / 1e3, mb, va, co, 4 * a['SQ_WAVE_CYCLES'] / c / (1024 * cyc)))
PY
