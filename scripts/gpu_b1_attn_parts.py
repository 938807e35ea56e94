"""MI355X: one 2048 x 2048 pair, fixed work, research library with the key range of the one-pair attention cut into `argv[1]` parts (4 = product, 8, 16): run under
rocprofv3 --kernel-trace --stats to see the attention and merge kernels' own times (is a walk of 8 tiles at 4 waves per SIMD faster than 16 tiles at 2?)."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); build = importlib.import_module('deep-image-matching_amd.build')
lib = capi.load(str(build.LIBDIR / "libdim_hip_research.so")); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
assert lib.dim_tune_set(12, 0 if parts == 4 else parts) == 0
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
g = torch.Generator().manual_seed(0)
kt = (torch.rand(2, 2048, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, 2048, 256, generator=g), dim=-1).cuda()
nt = torch.full((2,), 2048, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
m = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256, gain=2.0), conf, max_pairs=1, max_kpts=2048)
q = None
for _ in range(12): q = m.match_batch(kt, dt, nt, st, out=q)
torch.cuda.synchronize()
print(json.dumps({"parts": parts}))
