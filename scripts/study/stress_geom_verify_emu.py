"""CPU sweep (no GPU): the batched fundamental-matrix RANSAC (csrc/geom_verify.hip) on the test emulator against its numpy restatement
(oracle/geom_ref.py) and the synthetic ground truth on random two-view problems: 8 .. 150 matches, 0 .. 70 % outliers, noise 0 .. 1.5 px,
both error types.   python scripts/study/stress_geom_verify_emu.py SEED N"""
import ctypes, importlib, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import geom_ref
from tests.test_geom_verify_emu import _tables
build = importlib.import_module("deep-image-matching_amd.build")
verify = importlib.import_module("deep-image-matching_amd.verify")
lib = ctypes.CDLL(str(build.build_emu())); lib.dim_last_error.restype = ctypes.c_char_p
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for it in range(N):
    err = rnd.choice(["sampson", "symmetric_epipolar"])
    cases, truth = [], []
    for _ in range(3):
        tot = rnd.randint(8, 150); no = int(tot * rnd.choice([0.0, 0.2, 0.5, 0.7])); ni = max(8, tot - no)
        x0, x1, is_in, _ = geom_ref.synthetic_two_view(ni, no, seed=rnd.randrange(10000), noise_px=rnd.choice([0.0, 0.3, 1.5]), size=(640, 480))
        cases.append((x0, x1)); truth.append(is_in)
    thr, seed = rnd.choice([1.0, 1.5, 4.0]), rnd.randrange(100)
    try:
        kt, mt, n = _tables(cases, cap=256)
        v = verify.DeviceVerifier(threshold=thr, iters=256, error_type=err, seed=seed, device="cpu", lib=lib)
        out = v.verify_batch(kt, mt, n)
        for p, (x0, x1) in enumerate(cases):
            s = len(x0)
            F, mask, cnt, hid = geom_ref.fundamental_ransac(x0, x1, thr, iters=256, err_type=verify.ERROR_TYPES[err], seed=seed, pair=p)
            got = out["mask"][p, :s].numpy().astype(bool)
            assert int(out["n_inliers"][p]) == int(got.sum()) and not out["mask"][p, s:].any(), "mask bookkeeping"
            assert (got != mask).sum() <= 2 and abs(int(got.sum()) - cnt) <= 2, ("inlier sets", int((got != mask).sum()), int(got.sum()), cnt)
    except Exception as e:
        bad += 1; print("FAIL", it, dict(err=err, thr=thr, seed=seed, sizes=[len(c[0]) for c in cases]), repr(e)[:300], flush=True)
print("done", N, "failures", bad)
