"""Diagnostic: a batch of 8 ragged 2048-keypoint pairs vs the same pairs one by one, with the large-batch-only kernels forced on
the singles, for LayerNorm+GELU fused into ffn.0 (tune 11) off and on: max |log score| difference over the common matches."""
import importlib, sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lg = importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights = importlib.import_module('deep-image-matching_amd.weights')
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load()
sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
g = torch.Generator().manual_seed(4)
counts = [2048, 1900, 1777, 2048, 1500, 2001, 1999, 1024, 2048]
kt = torch.rand(9, 2048, 2, generator=g) * 1024
dt = torch.nn.functional.normalize(torch.randn(9, 2048, 256, generator=g), dim=-1)
nt, st = torch.tensor(counts, dtype=torch.int32), torch.tensor([[1024.0, 1024.0]] * 9)
pairs = torch.tensor([[i, i + 1] for i in range(8)], dtype=torch.int32)
res = {}
for mode in ("fixed", "default"):
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0} if mode == "fixed" else {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    for ln in (0, 1):
        lib.dim_tune_set(6, 2); lib.dim_tune_set(11, 2 if ln else 0)
        net = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
        singles = []
        for a, b in pairs.tolist():
            data = {"image0": {"keypoints": kt[a, :counts[a]][None], "descriptors": dt[a, :counts[a]][None], "image_size": st[a][None]},
                    "image1": {"keypoints": kt[b, :counts[b]][None], "descriptors": dt[b, :counts[b]][None], "image_size": st[b][None]}}
            r = net(data)
            singles.append((r["matches"][0].cpu(), r["scores"][0].cpu()))
        lib.dim_tune_set(6, 1); lib.dim_tune_set(11, 1 if ln else 0)
        big = lg.LightGlueHIP(sd, conf, max_pairs=8, max_kpts=2048)
        o = {k: v.cpu() for k, v in big.match_batch(kt.cuda(), dt.cuda(), nt.cuda(), st.cuda(), pair_idx=pairs.cuda()).items()}
        o2 = {k: v.cpu() for k, v in big.match_batch(kt.cuda(), dt.cuda(), nt.cuda(), st.cuda(), pair_idx=pairs.cuda()).items()}
        worst, same, rerun = 0.0, 0, bool(torch.equal(o["scores"], o2["scores"]))
        for p, (m, s) in enumerate(singles):
            S = int(o["n_matches"][p])
            eq = S == len(m) and bool(torch.equal(o["matches"][p, :S], m))
            same += eq
            if eq and S:
                worst = max(worst, float((o["scores"][p, :S].log() - s.log()).abs().max()))
        res[f"{mode}_ln{ln}"] = {"pairs_with_equal_matches": same, "max_abs_log_score_diff": worst, "batch_rerun_bit_equal": rerun}
lib.dim_tune_set(11, 3)
print(json.dumps(res, indent=1))
