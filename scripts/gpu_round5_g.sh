#!/bin/bash
# Round 5, call G: (1) the filler probe again with the MFMA accumulators in VGPRs (the form hipcc picks for every production kernel) next to AGPRs;
# (2) same-box A/B of library variants whose MFMA kernels are forced into the AGPR form per family (build.build_variant("agpr*")).
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 $R/scripts/probe/filler_probe 4000 > $O/${TAG}_filler_probe.jsonl 2> $O/${TAG}_filler_probe.err
tail -16 $O/${TAG}_filler_probe.jsonl
cd $R
L=$R/deep-image-matching_amd/lib
for V in product agprconv agprconv2 agprattn agprgemm agprall product2; do
  ARG=""; [ $V != product ] && [ $V != product2 ] && ARG="--lib $L/libdim_hip_$V.so"
  timeout 300 python bench.py $ARG --steps 10 --warmup 2 --no-cpu-baseline --no-strong-scaling --no-hook-path --main-region-only > $O/${TAG}_bench_$V.json 2>> $O/${TAG}_bench.err
  python -c "
import json,sys
d=json.loads(open('$O/${TAG}_bench_$V.json').read().strip().splitlines()[-1])
print('$V', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step conv1b', round(d['roofline']['avg_launch_ms'],2), 'clock', round(d['sustained_clock_mhz']), 'guard', d['fp16x3_range_guard']['violations'])"
done
cd /tmp
for V in product agprall; do
  ARG=""; [ $V != product ] && ARG="--lib $L/libdim_hip_$V.so"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$V -o bench -- python $R/bench.py $ARG --steps 3 --warmup 1 --no-cpu-baseline --no-strong-scaling --no-hook-path --main-region-only > /dev/null 2>&1
  f=$(find $O/prof_${TAG}_$V -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats_$V.csv
  rm -rf $O/prof_${TAG}_$V
done
