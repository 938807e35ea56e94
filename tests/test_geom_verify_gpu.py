"""GPU (MI355X): batched fundamental-matrix RANSAC (dim_gv_fundamental) at matching-batch sizes: 50 pairs x 2048 matches,
vs synthetic ground truth and vs the numpy oracle; retrieval top-k on the device."""
import importlib
import time

import numpy as np
import pytest
import torch

from oracle import geom_ref

pytestmark = pytest.mark.gpu


def test_batched_ransac_at_bench_batch_size(hip_lib):
    verify = importlib.import_module("deep-image-matching_amd.verify")
    P, S = 50, 2048
    kt = torch.zeros(2 * P, S, 2)
    mt = torch.zeros(P, S, 2, dtype=torch.int64)
    truth = []
    for p in range(P):
        ratio = 0.5 + 0.45 * (p / (P - 1))            # inlier ratios from 50 % to 95 % (0.5^7 x 4096 = 32 clean samples expected)
        ni = int(S * ratio)
        x0, x1, is_in, _ = geom_ref.synthetic_two_view(ni, S - ni, seed=p, noise_px=0.4)
        kt[2 * p], kt[2 * p + 1] = torch.from_numpy(x0), torch.from_numpy(x1)
        mt[p, :, 0] = mt[p, :, 1] = torch.arange(S)
        truth.append(is_in)
    n = torch.full((P,), S, dtype=torch.int32)
    n[3] = 5                                           # < 8 matches: everything is an inlier, F = 0 (reference rule)
    v = verify.DeviceVerifier(threshold=2.0, iters=4096, seed=3)
    kt_d, mt_d, n_d = kt.cuda(), mt.cuda(), n.cuda()
    out = v.verify_batch(kt_d, mt_d, n_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = v.verify_batch(kt_d, mt_d, n_d, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"dim_gv_fundamental: {P} pairs x {S} matches x 4096 hypotheses: {ms:.2f} ms per batch ({ms / P * 1e3:.0f} us per pair)")
    mask = out["mask"].cpu().numpy().astype(bool)
    ninl = out["n_inliers"].cpu().numpy()
    assert ninl[3] == 5 and mask[3, :5].all() and not mask[3, 5:].any() and float(out["F"][3].abs().sum()) == 0.0
    for p in range(P):
        if p == 3:
            continue
        tp, fp = (mask[p] & truth[p]).sum(), (mask[p] & ~truth[p]).sum()
        assert ninl[p] == mask[p].sum()
        assert tp >= 0.93 * truth[p].sum(), (p, tp, truth[p].sum())
        assert fp <= 0.12 * max(1, (~truth[p]).sum()), (p, fp)
        F = out["F"][p].cpu().numpy()
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-8
    again = v.verify_batch(kt_d, mt_d, n_d)
    assert torch.equal(again["mask"], out["mask"])     # deterministic
    assert ms < 50.0


def test_ransac_gpu_equals_numpy_oracle(hip_lib):
    verify = importlib.import_module("deep-image-matching_amd.verify")
    cases = [geom_ref.synthetic_two_view(300, 200, seed=11, noise_px=0.3)[:2], geom_ref.synthetic_two_view(150, 350, seed=12, noise_px=0.5)[:2]]
    S = 500
    kt = torch.zeros(4, S, 2); mt = torch.zeros(2, S, 2, dtype=torch.int64)
    for p, (x0, x1) in enumerate(cases):
        kt[2 * p], kt[2 * p + 1] = torch.from_numpy(x0), torch.from_numpy(x1)
        mt[p, :, 0] = mt[p, :, 1] = torch.arange(S)
    for err in ("sampson", "symmetric_epipolar"):
        v = verify.DeviceVerifier(threshold=1.5, iters=512, error_type=err, seed=5)
        out = v.verify_batch(kt.cuda(), mt.cuda(), torch.full((2,), S, dtype=torch.int32).cuda())
        for p, (x0, x1) in enumerate(cases):
            F, mask, cnt, _ = geom_ref.fundamental_ransac(x0, x1, 1.5, iters=512, err_type=verify.ERROR_TYPES[err], seed=5, pair=p)
            got = out["mask"][p].cpu().numpy().astype(bool)
            assert (got != mask).sum() <= 2 and abs(int(out["n_inliers"][p]) - cnt) <= 2


def test_retrieval_topk_gpu_vs_torch(hip_lib):
    pairs_mod = importlib.import_module("deep-image-matching_amd.pairs")
    g = torch.Generator().manual_seed(1)
    N, D, K = 1500, 4096, 20                             # NetVLAD-sized global descriptors
    desc = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1)
    names = [f"{i:05d}.jpg" for i in range(N)]
    got = pairs_mod.pairs_from_retrieval(names, names, desc.numpy(), desc.numpy(), num_matched=K)
    sim = desc.double() @ desc.double().t()
    sim.fill_diagonal_(float("-inf"))
    sim[sim < 0] = float("-inf")
    tk = torch.topk(sim, K, dim=1)
    # fp32 MFMA vs fp64 scores: the top-k SETS agree except for near-ties at the k-th place
    by_q = {}
    for a, b in got:
        by_q.setdefault(a, []).append(b)
    bad = 0
    for i in range(N):
        ref = {names[int(j)] for j, v in zip(tk.indices[i], tk.values[i]) if torch.isfinite(v)}
        bad += len(ref ^ set(by_q.get(names[i], [])))
    assert len(got) == N * K and bad <= 4


def test_device_ransac_vs_the_reference_estimator_iou(hip_lib):
    """VERDICT r2 next #1d: agreement of dim_gv_fundamental with the estimator the reference actually calls
    (cv2.findFundamentalMat(..., USAC_MAGSAC, gv_threshold, 0.9999, 10000), utils/geometric_verification.py:140-152) on synthetic
    two-view cases, as inlier-set IoU, driven through verify.HostVerifierPool (the reference's estimator on a thread pool).
    cv2 is not part of this image; the test runs wherever OpenCV is importable and is skipped (recorded) otherwise —
    the parity of f3 against cv2 therefore stays UNPINNED here."""
    import json
    from pathlib import Path
    verify = importlib.import_module("deep-image-matching_amd.verify")
    out_dir = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        import cv2  # noqa: F401
    except ImportError:
        try:
            out_dir.mkdir(exist_ok=True)
            with open(out_dir / "parity_measured.jsonl", "a") as f:
                f.write(json.dumps({"case": "device RANSAC vs cv2 USAC_MAGSAC", "status": "cv2 not importable on this box: unpinned"}) + "\n")
        except OSError:
            pass
        pytest.skip("cv2 is not importable on this box")
    cases = [geom_ref.synthetic_two_view(ni, no, seed=s, noise_px=0.4) for s, (ni, no) in enumerate([(1500, 500), (1000, 1000), (300, 100), (1800, 200)])]
    pool = verify.HostVerifierPool(method="MAGSAC", threshold=2.0, confidence=0.9999, max_iters=10000, workers=4)
    dv = verify.DeviceVerifier(threshold=2.0, iters=4096, seed=1)
    ious = []
    for x0, x1, is_in, _ in cases:
        m = np.stack([np.arange(len(x0)), np.arange(len(x0))], 1)
        _, ref_mask = pool.submit(x0, x1, m).result()
        _, got = dv.verify_pair(x0, x1, m)
        iou = (ref_mask & got).sum() / max(1, (ref_mask | got).sum())
        ious.append(float(iou))
    pool.shutdown()
    with open(out_dir / "parity_measured.jsonl", "a") as f:
        f.write(json.dumps({"case": "device RANSAC vs cv2 USAC_MAGSAC inlier IoU", "iou": ious}) + "\n")
    assert min(ious) >= 0.9, ious


def test_device_ransac_against_ground_truth_scenes(hip_lib):
    """f3 against GROUND TRUTH (VERDICT r3 weak #4 / next #8b): synthetic two-view scenes with a known F and a known inlier set
    (tests/two_view_truth.py — independent of geom_verify.hip and of oracle/geom_ref.py), 8 / 100 / 2048 matches x 0 - 70 % outliers:
    recall and precision of the device inlier mask, and the Sampson RMS of the CLEAN true correspondences under the returned F.
    cv2's USAC_MAGSAC (utils/geometric_verification.py:136-152) is not importable here, so parity with it stays unpinned; this is the
    contract it is called for.  Measured numbers go to gpurun_out/parity_measured.jsonl."""
    import json
    from pathlib import Path
    from tests import two_view_truth as tv
    verify = importlib.import_module("deep-image-matching_amd.verify")
    grid = [(8, 0.0), (8, 0.12), (100, 0.1), (100, 0.4), (100, 0.7), (2048, 0.1), (2048, 0.4), (2048, 0.7)]
    scenes = []
    for i, (n, rho) in enumerate(grid):
        no = int(round(n * rho))
        scenes.append(tv.scene(n - no, no, seed=50 + i))
    P, S = len(scenes), 2048
    kt = torch.zeros(2 * P, S, 2); mt = torch.zeros(P, S, 2, dtype=torch.int64); cnt = torch.zeros(P, dtype=torch.int32)
    for p, sc in enumerate(scenes):
        s = len(sc["x0"])
        kt[2 * p, :s], kt[2 * p + 1, :s] = torch.from_numpy(sc["x0"]), torch.from_numpy(sc["x1"])
        mt[p, :s, 0] = mt[p, :s, 1] = torch.arange(s)
        cnt[p] = s
    rows = []
    for iters in (4096, 32768):      # the reference's USAC budget is adaptive; 0.3^7 = 2.2e-4: 70 % outliers need the larger budget
        v = verify.DeviceVerifier(threshold=2.0, iters=iters, seed=9)
        out = v.verify_batch(kt.cuda(), mt.cuda(), cnt.cuda())
        mask, Fs = out["mask"].cpu().numpy().astype(bool), out["F"].cpu().numpy()
        for p, sc in enumerate(scenes):
            s = len(sc["x0"])
            r = tv.score(mask[p, :s], Fs[p], sc)
            rows.append({"test": "f3_ground_truth", "matches": grid[p][0], "outlier_ratio": grid[p][1], "iters": iters, **r})
            n, rho = grid[p]
            if n == 8 and rho > 0:       # 7 inliers + 1 outlier: only one all-inlier sample exists; report, assert the mask is sane
                assert mask[p, :s].sum() >= 7
                continue
            if rho >= 0.7 and iters < 32768:
                continue                 # reported, not asserted: too few hypotheses for a clean 7-sample
            assert r["recall"] >= 0.9 and r["precision"] >= (0.85 if rho >= 0.7 else 0.9), (grid[p], iters, r)
            assert r["sampson_rms_clean_px"] < (3.0 if n == 8 else 1.0), (grid[p], iters, r)
    d = Path(__file__).resolve().parents[1] / "gpurun_out"
    d.mkdir(exist_ok=True)
    with open(d / "parity_measured.jsonl", "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
