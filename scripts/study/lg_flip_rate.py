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


def margins(la):
    """top-2 margins of every row / column of the (m+1, n+1) log-assignment (dustbins excluded)"""
    r = torch.topk(la[:-1, :-1], 2, dim=1).values
    c = torch.topk(la[:-1, :-1], 2, dim=0).values
    return (r[:, 0] - r[:, 1]).float().numpy(), (c[0] - c[1]).float().numpy()


def run_cpu(n):
    from oracle import lightglue_ref
    res = {}
    for p in range(n):
        c = case_of(p)
        sd, f = gc.lg_weights(c), gc.lg_inputs(c)
        a = (f[0]["kpts"], f[0]["desc"], f[0]["size"], f[1]["kpts"], f[1]["desc"], f[1]["size"])
        r32 = lightglue_ref.lightglue_forward(*a, sd, CONF, taps=True)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        r64 = lightglue_ref.lightglue_forward(*a, sd64, {**CONF, "dtype": torch.float64}, taps=True)
        for tag, r in (("o32", r32), ("o64", r64)):
            res[f"{tag}_m_{p}"] = r["matches"].numpy().astype(np.int32)
            res[f"{tag}_s_{p}"] = r["scores"].float().numpy()
            res[f"{tag}_rm_{p}"], res[f"{tag}_cm_{p}"] = margins(r["log_assignment"])
        res[f"la_err_{p}"] = np.float32((r32["log_assignment"].double() - r64["log_assignment"]).abs().max().item())
        print(p, len(res[f"o32_m_{p}"]), len(res[f"o64_m_{p}"]), float(res[f"la_err_{p}"]), flush=True)
        if p % 20 == 19 or p == n - 1:
            np.savez_compressed(OUT / "lg_flip_oracle.npz", n=np.int32(p + 1), **res)


def run_gpu(n, batch=25):
    m = lambda name: importlib.import_module("deep-image-matching_amd." + name)
    dev = torch.device("cuda", 0)
    res = {}
    mats = {}
    for p0 in range(0, n, batch):
        ps = list(range(p0, min(n, p0 + batch)))
        for ws in sorted({case_of(p)["wseed"] for p in ps}):
            if ws not in mats:
                mats[ws] = m("lightglue_hip").LightGlueHIP(gc.lg_weights(case_of(ws - 1)), CONF, max_pairs=batch, max_kpts=NK, device=dev)
            sub = [p for p in ps if case_of(p)["wseed"] == ws]
            kt = torch.zeros(2 * len(sub), NK, 2); dt = torch.zeros(2 * len(sub), NK, 256)
            for q, p in enumerate(sub):
                f = gc.lg_inputs(case_of(p))
                kt[2 * q], kt[2 * q + 1], dt[2 * q], dt[2 * q + 1] = f[0]["kpts"], f[1]["kpts"], f[0]["desc"], f[1]["desc"]
            nt = torch.full((2 * len(sub),), NK, dtype=torch.int32)
            st = torch.full((2 * len(sub), 2), 1024.0)
            o = mats[ws].match_batch_guarded(kt.to(dev), dt.to(dev), nt.to(dev), st.to(dev), n_pairs=len(sub))
            for q, p in enumerate(sub):
                S = int(o["n_matches"][q])
                res[f"hip_m_{p}"] = o["matches"][q, :S].cpu().numpy().astype(np.int32)
                res[f"hip_s_{p}"] = o["scores"][q, :S].cpu().numpy()
        print("gpu pairs done:", ps[-1] + 1, flush=True)
    np.savez_compressed(OUT / "lg_flip_hip.npz", n=np.int32(n), **res)


def flips(ma, sa, mb, sb, th):
    a = {tuple(x) for x, s in zip(ma.tolist(), sa.tolist()) if s > th}
    b = {tuple(x) for x, s in zip(mb.tolist(), sb.tolist()) if s > th}
    return a, b, sorted(a ^ b)


def run_cmp():
    o = np.load(OUT / "lg_flip_oracle.npz")
    h = np.load(OUT / "lg_flip_hip.npz") if (OUT / "lg_flip_hip.npz").exists() else None
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
            row = {"study": "lg_flip_rate_2048", "pair": p, "threshold": th, "log_assignment_err_o32_vs_o64": float(o[f"la_err_{p}"])}
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
                if th == 0.0 and info:
                    tot[f"max_margin_{a}_vs_{b}"] = max(tot[f"max_margin_{a}_vs_{b}"], max(i["margin64"] for i in info))
            for k in ("o64", "o32", "hip"):
                if f"matches_{k}" in row:
                    tot[f"matches_{k}"] += row[f"matches_{k}"]
            rows.append(row)
        for a, b in (("o32", "o64"), ("hip", "o64"), ("hip", "o32")):
            tot[f"flip_rate_{a}_vs_{b}"] = tot[f"flips_{a}_vs_{b}"] / max(1, tot["matches_o64"])
        summ[f"threshold_{th}"] = tot
    prof = ROOT / "profiles"
    with open(prof / "r04_parity_measured.jsonl", "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    (prof / "r04_flip_rate_summary.json").write_text(json.dumps(summ, indent=1) + "\n")
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    OUT.mkdir(exist_ok=True)
    mode = sys.argv[1]
    if mode == "cpu":
        torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 6)
        run_cpu(int(sys.argv[2]))
    elif mode == "gpu":
        run_gpu(int(sys.argv[2]))
    else:
        run_cmp()
