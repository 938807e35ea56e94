"""MI355X: ALIKED (trained aliked-n16rot, config/aliked.yaml's values) on one 2000 x 2000 tile of real photographs — how far are the device's
score map and sub-pixel keypoints from the oracle's fp32 evaluation, and how far is THAT from an fp64 evaluation of the same module?
Writes gpurun_out/r06_aliked_tile_accuracy.json."""
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
torch.set_num_threads(16)
from oracle import aliked_ref  # noqa: E402
from tests import golden_cases as gc  # noqa: E402


def main():
    weights = importlib.import_module("deep-image-matching_amd.weights")
    al = importlib.import_module("deep-image-matching_amd.aliked_hip")
    cfg = {"model_name": "aliked-n16rot", "max_num_keypoints": 8000, "detection_threshold": 0.2, "nms_radius": 3}
    sd = weights.load_aliked_state_dict(str(ROOT / "tests/assets/aliked-n16rot.pth"), model_name="aliked-n16rot")
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    view = gc.real_mosaic(T + 48, T + 64)[:T, :T]
    img = torch.tensor(view.astype(np.float32).transpose(2, 0, 1)[None] / 255.0, dtype=torch.float)
    net = al.AlikedHIP(sd, cfg, max_batch=1, max_hw=(T, T), device="cuda")
    out = {k: v.cpu() for k, v in net(img).items()}
    taps = net.debug_taps()
    capi = importlib.import_module("deep-image-matching_amd.capi")
    from importlib import import_module
    sp = import_module("deep-image-matching_amd.superpoint_hip")
    score_dev = sp._copy_from(net.lib, taps["score_ptr"], (1, T, T), net.device)[0].cpu()
    r32 = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    r64 = aliked_ref.aliked_forward(img.double(), sd64, cfg, taps=True)
    s32, s64 = r32["score_map"][0, 0], r64["score_map"][0, 0]
    rec = {"tile": T, "score_map": {"dev_vs_ref32": float((score_dev - s32).abs().max()), "dev_vs_ref64": float((score_dev.double() - s64).abs().max()),
                                    "ref32_vs_ref64": float((s32.double() - s64).abs().max())}}
    from scipy.spatial import cKDTree

    def kp_diff(a, b):
        d, j = cKDTree(b.numpy().astype(np.float64)).query(a.numpy().astype(np.float64))
        ok = d <= 0.05
        per = (a.double()[ok] - b.double()[torch.from_numpy(j[ok])]).abs().max(1).values
        return {"paired": int(ok.sum()), "max": float(per.max()), "p99": float(per.quantile(0.99)), "median": float(per.median())}
    rec["keypoints_px"] = {"dev_vs_ref32": kp_diff(out["keypoints"], r32["keypoints"]), "dev_vs_ref64": kp_diff(out["keypoints"], r64["keypoints"].float()),
                           "ref32_vs_ref64": kp_diff(r32["keypoints"], r64["keypoints"].float())}
    print(json.dumps(rec))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r06_aliked_tile_accuracy.json").write_text(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
