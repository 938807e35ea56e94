import importlib, json, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
N = int(sys.argv[1]); cap = max(256, 1 << (N - 1).bit_length())
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
g = torch.Generator().manual_seed(0)
kt = (torch.rand(2, N, 2, generator=g) * 1024).cuda(); dt = torch.nn.functional.normalize(torch.randn(2, N, 256, generator=g), dim=-1).cuda()
nt = torch.full((2,), N, dtype=torch.int32, device='cuda'); st = torch.full((2, 2), 1024.0, device='cuda')
m = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap)
q = None
for _ in range(10): q = m.match_batch(kt, dt, nt, st, out=q)
torch.cuda.synchronize()
