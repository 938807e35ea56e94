"""MI355X (round 6): ONE pair per call with LightGlue's reference defaults (depth 0.95 / width 0.99), pairs designed to stop after 3 / 5 / 7 / 9 layers
(workloads.adaptive_lightglue_workload): ms per pair (HIP events + wall clock around the call incl. its synchronisation) with
  gated   : assignment launches after every layer, everything enqueued                     (dim_tune_set 17 = 0, 18 = 0: round 5)
  deferred: ONE assignment pass after the layer loop                                       (17 = 1, 18 = 0)
  followed: + the host follows the stop flags two layers behind and stops enqueueing layers (17 = 1, 18 = 1: the default)"""
import importlib, json, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load(); capi.install(lib, None)
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); wl = importlib.import_module('deep-image-matching_amd.workloads')
stops = (3, 5, 7, 9)
sd, kp, de, cnt, sz, expect = wl.adaptive_lightglue_workload(len(stops), stops=stops)
kp, de, cnt, sz = kp.cuda(), de.cuda(), cnt.cuda(), sz.cuda()
conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}
res = {}
for name, k17, k18 in (("gated", 0, 0), ("deferred", 1, 0), ("followed", 1, 1), ("gated again", 0, 0), ("followed again", 1, 1)):
    assert lib.dim_tune_set(17, k17) == 0 and lib.dim_tune_set(18, k18) == 0
    m = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
    rec = {}
    for p, s in enumerate(stops):
        pi = torch.tensor([[2 * p, 2 * p + 1]], dtype=torch.int32, device='cuda')
        q = [None]
        def f(): q[0] = m.match_batch(kp, de, cnt, sz, pair_idx=pi, out=q[0])
        for _ in range(5): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        rec[f"stop_{s}"] = {"wall_ms_per_call": round(wall, 4), "stop": int(q[0]["stop"][0]), "matches": int(q[0]["n_matches"][0])}
    res[name] = rec
    del m
lib.dim_tune_set(17, 1); lib.dim_tune_set(18, 1)
print(json.dumps(res))
