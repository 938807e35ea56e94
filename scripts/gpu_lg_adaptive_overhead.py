"""GPU: what the adaptive machinery of LightGlue (token confidence + early-stop decision + pruning, LGN:586-604; reference defaults depth 0.95 / width 0.99) costs
at the headline batch (50 pairs x 2048 x 2048 keypoints, 9 layers) when it changes nothing — seeded synthetic weights never stop early and never prune, so the
difference to the fixed-work call is pure overhead.  pruning_min_kpts -1: prune at every size (the reference's CPU behaviour; its GPU default 1536 would prune too)."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
dev = torch.device("cuda:0"); P, N = 50, 2048
g = torch.Generator().manual_seed(3)
kt = (torch.rand(2 * P, N, 2, generator=g) * 1000).to(dev); dt = torch.nn.functional.normalize(torch.randn(2 * P, N, 256, generator=g), dim=-1).to(dev)
nt = torch.full((2 * P,), N, dtype=torch.int32, device=dev); st = torch.full((2 * P, 2), 1024.0, device=dev)
sd = weights.synthetic_lightglue_state_dict(0, 256)
res = {}
for name, conf in (("fixed_work", {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}),
                   ("reference_default", {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": -1})):
    net = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N, device=dev)
    out = net.match_batch(kt, dt, nt, st)
    for _ in range(2): net.match_batch(kt, dt, nt, st, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): net.match_batch(kt, dt, nt, st, out=out)
    e1.record(); torch.cuda.synchronize()
    res[name] = {"ms_per_50_pairs": round(e0.elapsed_time(e1) / 5, 3), "stop_layers": sorted(set(out["stop"].cpu().tolist())), "matches_mean": float(out["n_matches"].float().mean())}
    del net
res["overhead_ms"] = round(res["reference_default"]["ms_per_50_pairs"] - res["fixed_work"]["ms_per_50_pairs"], 3)
print(json.dumps(res))
