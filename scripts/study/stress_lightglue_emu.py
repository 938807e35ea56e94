"""CPU sweep (no GPU): random LightGlue configurations — 0 .. 90 keypoints per image (empty and single-keypoint images included), 256- and 128-d
descriptors, adaptive depth / width on and off, correlated or random descriptors, and both kernel families (small-batch kernels / the forced
large-batch ones: K|V-image projection blocks, one-kernel feed-forward) — through the HIP sources on the test emulator against the oracle with
compare_lightglue (near-tie rule on).   python scripts/study/stress_lightglue_emu.py SEED N"""
import importlib, sys, random, torch, ctypes
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import lightglue_ref
from tests import golden_cases as gc
from tests.parity import compare_lightglue
build = importlib.import_module("deep-image-matching_amd.build")
lg_mod = importlib.import_module("deep-image-matching_amd.lightglue_hip")
weights = importlib.import_module("deep-image-matching_amd.weights")
lib = ctypes.CDLL(str(build.build_emu())); lib.dim_last_error.restype = ctypes.c_char_p
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(N):
    m, n = rnd.choice([0, 1, 2, 5, 17, 33, 64, 70]), rnd.choice([0, 1, 3, 8, 31, 32, 65, 90])
    dim = rnd.choice([256, 128])
    conf = {"depth_confidence": rnd.choice([-1, 0.95, 0.5]), "width_confidence": rnd.choice([-1, 0.99, 0.8]), "filter_threshold": rnd.choice([0.0, 0.1]),
            "pruning_min_kpts": -1, "n_layers": rnd.choice([2, 3])}
    wseed = rnd.randrange(100)
    sd = weights.synthetic_lightglue_state_dict(wseed, dim, n_layers=conf["n_layers"], gain=2.0)
    hg = rnd.choice([1.0, 30.0])
    for k in sd:
        if k.endswith("weight") and ("matchability" in k or "token_confidence" in k): sd[k] = sd[k] * hg
    case = {"seed": rnd.randrange(1000), "m": m, "n": n, "input_dim": dim, "size0": (480.0, 640.0), "size1": (512.0, 384.0)}
    g = torch.Generator().manual_seed(case["seed"])
    def feats(k, size):
        kp = torch.rand(k, 2, generator=g) * torch.tensor([size[1], size[0]])
        de = torch.nn.functional.normalize(torch.randn(k, dim, generator=g), dim=-1) if k else torch.zeros(0, dim)
        return {"kpts": kp, "desc": de, "size": torch.tensor(size)}
    f0, f1 = feats(m, case["size0"]), feats(n, case["size1"])
    if m and n and rnd.random() < 0.5:      # correlated descriptors: real matches
        k = min(m, n); f1["desc"][:k] = torch.nn.functional.normalize(f0["desc"][:k] + 0.3 * torch.randn(k, dim, generator=g), dim=-1)
    big = rnd.random() < 0.4
    lib.dim_tune_set(6, 2 if big else 1); lib.dim_tune_set(11, 4 if big else 3)
    try:
        net = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=max(m, n, 4), device="cpu", lib=lib)
        data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
                "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
        out = net(data, dense=True)
        out = {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in out.items()}
        ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
        res = compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"], filter_threshold=conf["filter_threshold"])
        if res.get("explained_near_ties"): print("tie", it, m, n, res["explained_near_ties"])
    except Exception as e:
        bad += 1; print("FAIL", it, dict(m=m, n=n, dim=dim, conf=conf, wseed=wseed, case=case, big=big), repr(e)[:300], flush=True)
lib.dim_tune_set(6, 1); lib.dim_tune_set(11, 3)
print("done", N, "failures", bad)
