"""Match-list flip rate of LightGlue at the BASELINE size (2048 x 2048 keypoints) — VERDICT r3 next #2.

A "flip" = a match (i, j) that one evaluation reports and another does not.  Three evaluations of the same network on the same
seeded inputs (tests/golden_cases.lg_inputs: image 1 is a perturbed, shuffled copy of image 0, so true correspondences exist):

  hip    the product path on the MI355X (default fp16x3 arithmetic), fixed work (depth / width -1), threshold 0
  o32    the oracle in fp32  (= the reference's arithmetic; oracle/lightglue_ref.py is pinned to the reference module)
  o64    the oracle in fp64  (same network, double precision: the yardstick)

The fp32-vs-fp64 flip rate of the ORACLE is the rate at which the reference itself is not reproducible; the product is held to
that rate (tests/test_lightglue_gpu.py::test_match_list_flip_rate_at_2048 reads profiles/r04_flip_rate_summary.json).
The threshold-0.1 lists are the threshold-0 lists restricted to score > 0.1 (filter_matches, LGN:41-58), so one run gives both.

    python scripts/study/lg_flip_rate.py gpu  N   -> gpurun_out/lg_flip_hip.npz          (on the GPU box)
    python scripts/study/lg_flip_rate.py cpu  N   -> gpurun_out/lg_flip_oracle.npz       (anywhere; ~6 s per pair on 8 cores)
    python scripts/study/lg_flip_rate.py cmp      -> profiles/r04_parity_measured.jsonl (appended) + r04_flip_rate_summary.json

Round 5 — the same study WHERE THE REFERENCE OPERATES: threshold 0.1 with hundreds of matches per pair, adaptive depth and
width ON (0.95 / 0.99).  Inputs of graded difficulty (``case2_of``): image 1's descriptors are image 0's plus noise of a per-keypoint level
drawn log-uniformly from [0.3, 6] x the descriptor norm, a quarter of them replaced by unrelated descriptors; matching-capable weights of
moderate sharpness (weights.synthetic_lightglue_matching_state_dict, sharpness 40), so the matching scores cover (0, 1) and many decisions sit
near the threshold and near ties:

    python scripts/study/lg_flip_rate.py cpu2 N [threads] -> gpurun_out/lg_flip2_oracle.npz  (build container: ~5 s per pair on 8 cores)
    python scripts/study/lg_flip_rate.py gpu2 N           -> gpurun_out/lg_flip2_hip.npz     (GPU box: the HIP path only, seconds)
    python scripts/study/lg_flip_rate.py cmp2             -> profiles/r05_flip_rate_summary.json + r05_parity_measured.jsonl (appended)
"""
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import golden_cases as gc  # noqa: E402

OUT = ROOT / "gpurun_out"
CONF = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
NK = 2048


def case_of(p):
    c = dict(gc.LG_CASES["fixed"])
    c.update(m=NK, n=NK, seed=5000 + p, wseed=1 + p % 4, size0=(1024.0, 1024.0), size1=(1024.0, 1024.0))
    return c


CONF2 = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}


def case2_of(p):
    """(weights, (f0, f1)) of pair p of the round-5 study: 2048 x 2048 keypoints, graded difficulty, a quarter of image 1 without a partner."""
    import math
    weights = importlib.import_module("deep-image-matching_amd.weights")
    g = torch.Generator().manual_seed(7000 + p)
    D = 256
    k0 = torch.rand(NK, 2, generator=g) * 1024.0
    d0 = torch.nn.functional.normalize(torch.randn(NK, D, generator=g), dim=-1)
    perm = torch.randperm(NK, generator=g)
    k1 = k0[perm] + torch.randn(NK, 2, generator=g) * 2.0
    sigma = torch.exp(torch.rand(NK, 1, generator=g) * (math.log(6.0) - math.log(0.3)) + math.log(0.3))
    d1 = d0[perm] + sigma * torch.randn(NK, D, generator=g) / math.sqrt(D)
    lone = torch.rand(NK, generator=g) < 0.25
    d1[lone] = torch.randn(int(lone.sum()), D, generator=g)
    d1 = torch.nn.functional.normalize(d1, dim=-1)
    size = torch.tensor([1024.0, 1024.0])
    sd = weights.synthetic_lightglue_matching_state_dict(1 + p % 4, 256, sharpness=40.0)
    if p % 2 == 1:     # every second pair has confident token heads: the depth criterion (LGN:593-604) stops it after 3 - 4 layers
        for k in sd:
            if k.startswith("token_confidence") and k.endswith("bias"):
                sd[k] = sd[k] + 1.7
    return sd, ({"kpts": k0.contiguous(), "desc": d0.contiguous(), "size": size}, {"kpts": k1.contiguous(), "desc": d1.contiguous(), "size": size})


def margins(la):
    """top-2 margins of every row / column of the (m+1, n+1) log-assignment (dustbins excluded)"""
    r = torch.topk(la[:-1, :-1], 2, dim=1).values
    c = torch.topk(la[:-1, :-1], 2, dim=0).values
    return (r[:, 0] - r[:, 1]).float().numpy(), (c[0] - c[1]).float().numpy()


def full_margins(r, n_rows=NK, n_cols=NK):
    """row / column top-2 margins scattered back to the ORIGINAL keypoint indices (with pruning the log-assignment lives in the pruned index
    space: ind0 / ind1); +inf where a keypoint was pruned (no decision to flip there)"""
    rm, cm = margins(r["log_assignment"])
    fr, fc = np.full(n_rows, np.inf, np.float32), np.full(n_cols, np.inf, np.float32)
    i0 = r["ind0"].numpy() if r.get("ind0") is not None else np.arange(len(rm))
    i1 = r["ind1"].numpy() if r.get("ind1") is not None else np.arange(len(cm))
    fr[i0], fc[i1] = rm, cm
    return fr, fc


def run_cpu(n, v2=False):
    from oracle import lightglue_ref
    res = {}
    conf = CONF2 if v2 else CONF
    for p in range(n):
        if v2:
            sd, f = case2_of(p)
        else:
            c = case_of(p)
            sd, f = gc.lg_weights(c), gc.lg_inputs(c)
        a = (f[0]["kpts"], f[0]["desc"], f[0]["size"], f[1]["kpts"], f[1]["desc"], f[1]["size"])
        r32 = lightglue_ref.lightglue_forward(*a, sd, conf, taps=True)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        r64 = lightglue_ref.lightglue_forward(*a, sd64, {**conf, "dtype": torch.float64}, taps=True)
        for tag, r in (("o32", r32), ("o64", r64)):
            res[f"{tag}_m_{p}"] = r["matches"].numpy().astype(np.int32)
            res[f"{tag}_s_{p}"] = r["scores"].float().numpy()
            res[f"{tag}_rm_{p}"], res[f"{tag}_cm_{p}"] = full_margins(r) if v2 else margins(r["log_assignment"])
            res[f"{tag}_stop_{p}"] = np.int32(r["stop"])
        same_shape = r32["log_assignment"].shape == r64["log_assignment"].shape
        res[f"la_err_{p}"] = np.float32((r32["log_assignment"].double() - r64["log_assignment"]).abs().max().item()) if same_shape else np.float32(np.nan)
        print(p, len(res[f"o32_m_{p}"]), len(res[f"o64_m_{p}"]), int((res[f"o32_s_{p}"] > 0.1).sum()), "stop", int(r32["stop"]), int(r64["stop"]),
              float(res[f"la_err_{p}"]), flush=True)
        if p % 20 == 19 or p == n - 1:
            np.savez_compressed(OUT / ("lg_flip2_oracle.npz" if v2 else "lg_flip_oracle.npz"), n=np.int32(p + 1), **res)


def run_gpu(n, batch=25, v2=False):
    m = lambda name: importlib.import_module("deep-image-matching_amd." + name)
    dev = torch.device("cuda", 0)
    res = {}
    mats = {}
    wseed = (lambda p: 1 + p % 4) if v2 else (lambda p: case_of(p)["wseed"])
    for p0 in range(0, n, batch):
        ps = list(range(p0, min(n, p0 + batch)))
        for ws in sorted({wseed(p) for p in ps}):
            if ws not in mats:
                sd_ws = case2_of(ws - 1)[0] if v2 else gc.lg_weights(case_of(ws - 1))
                mats[ws] = m("lightglue_hip").LightGlueHIP(sd_ws, {**(CONF2 if v2 else CONF), "pruning_min_kpts": -1}, max_pairs=batch, max_kpts=NK, device=dev)
            sub = [p for p in ps if wseed(p) == ws]
            kt = torch.zeros(2 * len(sub), NK, 2); dt = torch.zeros(2 * len(sub), NK, 256)
            for q, p in enumerate(sub):
                f = case2_of(p)[1] if v2 else gc.lg_inputs(case_of(p))
                kt[2 * q], kt[2 * q + 1], dt[2 * q], dt[2 * q + 1] = f[0]["kpts"], f[1]["kpts"], f[0]["desc"], f[1]["desc"]
            nt = torch.full((2 * len(sub),), NK, dtype=torch.int32)
            st = torch.full((2 * len(sub), 2), 1024.0)
            o = mats[ws].match_batch_guarded(kt.to(dev), dt.to(dev), nt.to(dev), st.to(dev), n_pairs=len(sub))
            for q, p in enumerate(sub):
                S = int(o["n_matches"][q])
                res[f"hip_m_{p}"] = o["matches"][q, :S].cpu().numpy().astype(np.int32)
                res[f"hip_s_{p}"] = o["scores"][q, :S].cpu().numpy()
                res[f"hip_stop_{p}"] = np.int32(int(o["stop"][q]))
        print("gpu pairs done:", ps[-1] + 1, flush=True)
    np.savez_compressed(OUT / ("lg_flip2_hip.npz" if v2 else "lg_flip_hip.npz"), n=np.int32(n), **res)


def flips(ma, sa, mb, sb, th):
    a = {tuple(x) for x, s in zip(ma.tolist(), sa.tolist()) if s > th}
    b = {tuple(x) for x, s in zip(mb.tolist(), sb.tolist()) if s > th}
    return a, b, sorted(a ^ b)


def run_cmp(v2=False):
    tagf = "lg_flip2" if v2 else "lg_flip"
    o = np.load(OUT / f"{tagf}_oracle.npz")
    h = np.load(OUT / f"{tagf}_hip.npz") if (OUT / f"{tagf}_hip.npz").exists() else None
    n = int(o["n"]) if h is None else min(int(o["n"]), int(h["n"]))
    rows, summ = [], {}
    for th in (0.0, 0.1):
        tot = {"pairs": n, "threshold": th, "matches_o64": 0, "matches_o32": 0, "matches_hip": 0, "flips_o32_vs_o64": 0, "flips_hip_vs_o64": 0,
               "flips_hip_vs_o32": 0, "max_margin_o32_vs_o64": 0.0, "max_margin_hip_vs_o32": 0.0, "max_margin_hip_vs_o64": 0.0,
               "pairs_with_flip_o32_vs_o64": 0, "pairs_with_flip_hip_vs_o32": 0, "pairs_with_flip_hip_vs_o64": 0}
        for p in range(n):
            ev = {"o32": (o[f"o32_m_{p}"], o[f"o32_s_{p}"]), "o64": (o[f"o64_m_{p}"], o[f"o64_s_{p}"])}
            if h is not None:
                ev["hip"] = (h[f"hip_m_{p}"], h[f"hip_s_{p}"])
            rm, cm = o[f"o64_rm_{p}"], o[f"o64_cm_{p}"]     # decision margins in the fp64 evaluation
            row = {"study": "lg_flip_rate_2048_default_threshold_adaptive" if v2 else "lg_flip_rate_2048", "pair": p, "threshold": th, "log_assignment_err_o32_vs_o64": float(o[f"la_err_{p}"])}
            for a, b in (("o32", "o64"), ("hip", "o64"), ("hip", "o32")):
                if a not in ev:
                    continue
                sa, sb, d = flips(*ev[a], *ev[b], th)
                row[f"matches_{a}"], row[f"matches_{b}"] = len(sa), len(sb)
                # a flip near the threshold has a large assignment margin but a score within noise of th: report both
                info = [{"match": list(x), "margin64": float(min(rm[x[0]], cm[x[1]]))} for x in d]
                row[f"flips_{a}_vs_{b}"] = info
                tot[f"flips_{a}_vs_{b}"] += len(d)
                tot[f"pairs_with_flip_{a}_vs_{b}"] += 1 if d else 0
                if info and (th == 0.0 or v2):
                    fin = [i["margin64"] for i in info if np.isfinite(i["margin64"])]
                    if fin:
                        tot[f"max_margin_{a}_vs_{b}"] = max(tot[f"max_margin_{a}_vs_{b}"], max(fin))
                if v2 and th == 0.1 and d:       # a flip at the default threshold: a score within noise of 0.1, or an assignment tie
                    sc = {tuple(x): float(s_) for (mm_, ss_) in (ev[a], ev[b]) for x, s_ in zip(mm_.tolist(), ss_.tolist())}
                    tot.setdefault(f"flips_near_threshold_{a}_vs_{b}", 0)
                    tot[f"flips_near_threshold_{a}_vs_{b}"] += sum(1 for x in d if abs(sc.get(tuple(x), 0.0) - 0.1) < 1e-3)
            for k in ("o64", "o32", "hip"):
                if f"matches_{k}" in row:
                    tot[f"matches_{k}"] += row[f"matches_{k}"]
            rows.append(row)
        for a, b in (("o32", "o64"), ("hip", "o64"), ("hip", "o32")):
            tot[f"flip_rate_{a}_vs_{b}"] = tot[f"flips_{a}_vs_{b}"] / max(1, tot["matches_o64"])
        summ[f"threshold_{th}"] = tot
    prof = ROOT / "profiles"
    if v2 and h is not None:
        summ["stop_layer_differs_hip_vs_o32"] = int(sum(int(h[f"hip_stop_{p}"]) != int(o[f"o32_stop_{p}"]) for p in range(n)))
    if v2:
        summ["stop_layer_differs_o32_vs_o64"] = int(sum(int(o[f"o32_stop_{p}"]) != int(o[f"o64_stop_{p}"]) for p in range(n)))
    rnd = "r05" if v2 else "r04"
    with open(prof / f"{rnd}_parity_measured.jsonl", "a") as f:
        for r in rows:
            if v2 and not any(r.get(k) for k in r if k.startswith("flips_")):
                continue          # round 5: only the pairs that carry a flip are listed (the totals are in the summary)
            f.write(json.dumps(r) + "\n")
    (prof / f"{rnd}_flip_rate_summary.json").write_text(json.dumps(summ, indent=1) + "\n")
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    OUT.mkdir(exist_ok=True)
    mode = sys.argv[1]
    if mode in ("cpu", "cpu2"):
        torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 6)
        run_cpu(int(sys.argv[2]), v2=mode == "cpu2")
    elif mode in ("gpu", "gpu2"):
        run_gpu(int(sys.argv[2]), v2=mode == "gpu2")
    else:
        run_cmp(v2=mode == "cmp2")
