"""Phase timers of the Winograd conv1b kernel (dim_tune_set(15, 1 | 8) fused staging, (15, 3 | 8) two-phase): average cycles per phase
per workgroup (wave 0), from s_memtime stamps inside the kernel.  python scripts/gpu_wino_phases.py"""
import ctypes, importlib, json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
capi = importlib.import_module("deep-image-matching_amd.capi")
sp = importlib.import_module("deep-image-matching_amd.superpoint_hip")
weights = importlib.import_module("deep-image-matching_amd.weights")
lib = capi.load(str(capi.LIB_PATH.parent / 'libdim_hip_research.so')); capi.install(lib, None)   # research build: dim_tune_set keys 12-15 (timing probes / prototypes) exist only there
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
net = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=100, max_hw=(1024, 1024), capacity=2048)
imgs = torch.rand(100, 1024, 1024, generator=torch.Generator().manual_seed(0)).cuda()
names = ["prologue", "staging (fused) / phase 2", "barrier after staging", "MFMA steps", "phase 1 of next chunk", "barrier after MFMA interval",
         "output transform (exchange)", "epilogue", "whole workgroup", "workgroups"]
for var, label in ((1 | 8, "fused staging"), (3 | 8, "two-phase staging")):
    lib.dim_tune_set(15, var)
    net.extract_batch(imgs); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.dim_conv_wg_phase_read(buf, 1)
    net.extract_batch(imgs); net.extract_batch(imgs); torch.cuda.synchronize()
    lib.dim_conv_wg_phase_read(buf, 1)
    n = max(1, buf[9])
    print(json.dumps({"variant": label, "workgroups": int(buf[9]), **{names[i]: round(buf[i] / n, 1) for i in range(9)}}))
lib.dim_tune_set(15, 0)
