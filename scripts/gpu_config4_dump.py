"""GPU half of tests/test_configs_gpu.py::test_config4_exhaustive_pairs_through_the_pipeline_vs_oracle, dumped for an offline
comparison with the oracle (the 276 oracle runs are CPU time that need not be spent on the GPU box): gpurun_out/config4_dump.pt."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_m = lambda n: importlib.import_module("deep-image-matching_amd." + n)
weights, pl = _m("weights"), _m("pipeline")
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 400, "remove_borders": 4}
conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0, "pruning_min_kpts": -1}
sp_sd, lg_sd = weights.synthetic_superpoint_state_dict(1234), weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
n_img, H, W = 24, 200, 264
imgs = torch.rand(n_img, H, W, generator=torch.Generator().manual_seed(4))
ext = _m("superpoint_hip").SuperPointHIP(sp_sd, cfg, max_batch=8, max_hw=(H, W), capacity=400)
mat = _m("lightglue_hip").LightGlueHIP(lg_sd, conf, max_pairs=16, max_kpts=400)
pipe = pl.PairMatchingPipeline(ext, mat)
table = pipe.extract_all(imgs.cuda())
pairs = pl.exhaustive_pairs(n_img)
out = [t.cpu() for t in pipe.match_all(table, pairs, aux=True)]
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"table": [t.cpu() for t in table], "pairs": pairs.cpu(), "out": out}, "gpurun_out/config4_dump.pt")
print("dumped", [tuple(t.shape) for t in out])
