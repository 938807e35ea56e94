"""CPU sweep (no GPU): random ALIKED configurations — all four geometries (t16 / n16 / n16rot / n32), sides that are not multiples of 32 (replicate padding), gray / RGB, NMS radius 2-3,
n_limit binding or not, both arithmetics (fp16x3 matrix-core path / fp32 paths) — through the HIP sources on the test emulator against the
oracle with compare_aliked (1e-3, exact keypoint set up to explained ties).   python scripts/study/stress_aliked_emu.py SEED N"""
import ctypes, importlib, os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import aliked_ref
from tests.test_aliked_emu import compare_aliked
build = importlib.import_module("deep-image-matching_amd.build")
al_mod = importlib.import_module("deep-image-matching_amd.aliked_hip")
weights = importlib.import_module("deep-image-matching_amd.weights")
lib = ctypes.CDLL(str(build.build_emu())); lib.dim_last_error.restype = ctypes.c_char_p
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for it in range(N):
    H, W, C = rnd.randint(40, 100), rnd.randint(40, 120), rnd.choice([1, 3])
    model = rnd.choice(["aliked-n16rot", "aliked-n32", "aliked-t16", "aliked-n16"])
    cfg = {"model_name": model, "max_num_keypoints": rnd.choice([30, 200, 4000]), "detection_threshold": rnd.choice([0.2, 0.1, 0.4]),
           "nms_radius": rnd.choice([2, 3])}
    sd = weights.synthetic_aliked_state_dict(rnd.randrange(40), model)
    img = torch.rand(1, C, H, W, generator=torch.Generator().manual_seed(rnd.randrange(10000)))
    arith = rnd.choice([2, 2, 0])
    lib.dim_tune_set(1, arith)
    try:
        net = al_mod.AlikedHIP(sd, cfg, max_batch=1, max_hw=(H, W), capacity=4096, device="cpu", lib=lib)
        out = {k: v.cpu() for k, v in net(img).items()}
        ref = aliked_ref.aliked_forward(img, sd, cfg, taps=True)
        compare_aliked(out, ref, ref_score_map=ref.get("score_map"), threshold=cfg["detection_threshold"], nms_radius=cfg["nms_radius"],
                       n_limit=cfg["max_num_keypoints"])
    except Exception as e:
        bad += 1; print("FAIL", it, dict(H=H, W=W, C=C, cfg=cfg, arith=arith), repr(e)[:400], flush=True)
    finally:
        lib.dim_tune_set(1, 2)
print("done", N, "failures", bad)
