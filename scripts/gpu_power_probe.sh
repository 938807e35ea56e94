#!/bin/bash
# sustained matrix-core rate under the chip's power management + rocm-smi power / clock samples taken while the probe runs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
rocm-smi --showmaxpower --showpower --showclocks > $O/power_probe_smi_idle.txt 2>&1
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/power_probe_smi_samples.txt &
SMI=$!
./scripts/probe/mfma_power_probe 2048 1200 | tee $O/power_probe.jsonl
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
sort $O/power_probe_smi_samples.txt | uniq -c | sort -rn | head -12
