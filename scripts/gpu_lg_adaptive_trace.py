"""MI355X: only the reference-default (adaptive depth 0.95 / width 0.99) LightGlue call of bench.measure_adaptive, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_adaptive -- python scripts/gpu_lg_adaptive_trace.py [fixed]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
wl = importlib.import_module("deep-image-matching_amd.workloads")
dev = torch.device("cuda:0")
P, N = 50, 2048
sd, kp, de, nt, st, expect = wl.adaptive_lightglue_workload(P, N)
kp, de, nt, st = kp.to(dev), de.to(dev), nt.to(dev), st.to(dev)
fixed = len(sys.argv) > 1 and sys.argv[1] == "fixed"
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1} if fixed else \
       {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": 1536}
net = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N, device=dev)
out = net.match_batch(kp, de, nt, st)
for _ in range(5):
    net.match_batch(kp, de, nt, st, out=out)
torch.cuda.synchronize()
print("stop", sorted(set(out["stop"].cpu().tolist())))
