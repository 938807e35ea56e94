"""One ALIKED extraction workload for rocprofv3: B tiles of H x W RGB, `reps` calls (first call = warm-up).
    python scripts/gpu_aliked_one.py [B=8] [H=1000] [W=1500] [reps=3]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, H, W, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 1000), (3, 1500), (4, 3)))
al = importlib.import_module('deep-image-matching_amd.aliked_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 4000, "detection_threshold": 0.2, "nms_radius": 3}
net = al.AlikedHIP(weights.synthetic_aliked_state_dict(7), cfg, max_batch=B, max_hw=(H, W), capacity=4000)
imgs = torch.rand(B, H, W, 3, device='cuda')
for _ in range(reps + 1):
    out = net.extract_batch(imgs)
torch.cuda.synchronize()
print("n", out[3].tolist())
