#!/bin/bash
# GPU A/B of two builds of the library in ONE call (boxes differ in sustained clock): bench.py on the default build and on
# deep-image-matching_amd/lib/libdim_hip_B.so, alternating.  usage: gpu_ab_libs.sh [rounds]
R=${1:-2}
for r in $(seq $R); do
  for L in A B; do
    if [ $L = B ]; then ARG="--lib deep-image-matching_amd/lib/libdim_hip_B.so"; else ARG=""; fi
    python bench.py --no-cpu-baseline --main-region-only $ARG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), round(d['sustained_clock_mhz']), round(d['roofline']['avg_launch_ms'],2))"
  done
done
