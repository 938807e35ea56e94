#!/bin/bash
# Round 4, GPU call B: Winograd F(2,3) conv1b (dim_tune_set(15, 1)) vs the direct kernel on ONE box: parity test, bench A/B (twice each),
# rocprofv3 kernel stats + MFMA / LDS counters of both variants.
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_superpoint_gpu.py -m gpu -q -x -k winograd > $O/${TAG}_wino_test.log 2>&1; tail -3 $O/${TAG}_wino_test.log
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-strong-scaling > $O/${TAG}_bench_direct$rep.json 2>> $O/${TAG}_bench.err
  python bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=1 --no-cpu-baseline --no-strong-scaling > $O/${TAG}_bench_wino$rep.json 2>> $O/${TAG}_bench.err
done
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_w$V -o bench -- python $R/bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=$V --steps 3 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${TAG}_w${V}_MFMA -o pmc -- python $R/bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=$V --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_${TAG}_w${V}_LDS -o pmc -- python $R/bench.py --lib $R/deep-image-matching_amd/lib/libdim_hip_research.so --tune 15=$V --steps 2 --warmup 1 --no-cpu-baseline --main-region-only > /dev/null 2>&1
done
cd $R
python - <<'PY'
import json, glob, os, csv, collections, re
O = "gpurun_out"; TAG = os.environ.get("TAG", "r04b")
for f in sorted(glob.glob(f"{O}/{TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 2), "ms/step clock", round(d["sustained_clock_mhz"]), "conv1b ms", round(d["roofline"]["avg_launch_ms"], 2), "guard", d["fp16x3_range_guard"]["violations"])
    except Exception as e:
        print(f, "ERR", e)
for V in (0, 1):
    for kind in ("MFMA", "LDS"):
        p = f"{O}/pmc_{TAG}_w{V}_{kind}"
        fs = glob.glob(p + "/**/*counter_collection.csv", recursive=True)
        if not fs: print(p, "no csv"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
        for r in csv.DictReader(open(fs[0])):
            n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:60]
            if "conv3x3" not in n: continue
            n += " g" + str(int(r["Grid_Size"]) // 256)
            agg[n][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        for n in sorted(agg, key=lambda n: -dur[n])[:2]:
            c = cnt[n]; print(f"w{V} {kind} {n}: calls {c} avg_us {dur[n]/c/1e3:.1f} " + " ".join(f"{k}={v/c:.4g}" for k, v in agg[n].items()))
PY
