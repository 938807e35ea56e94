"""GPU (MI355X): LightGlue HIP path through the C ABI vs the oracle / reference goldens."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lightglue_ref
from tests import golden_cases as gc
from tests.parity import compare_lightglue

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _lg():
    return importlib.import_module("deep-image-matching_amd.lightglue_hip")


def _cpu(out):
    return {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in out.items()}


def _data(f0, f1):
    return {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}


@pytest.mark.parametrize("name", list(gc.LG_CASES))
def test_lightglue_gpu_vs_reference_golden(hip_lib, name):
    case = gc.LG_CASES[name]
    sd = gc.lg_weights(case)
    f0, f1 = gc.lg_inputs(case)
    net = _lg().LightGlueHIP(sd, case["conf"], max_pairs=1, max_kpts=max(case["m"], case["n"]))
    out = _cpu(net(_data(f0, f1), dense=True))
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, case["conf"], taps=True)
    compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    g = np.load(GOLD / f"lg_{name}.npz")
    gold = {k: torch.from_numpy(np.asarray(g[k])) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1",
                                                             "matches", "scores", "prune0", "prune1")}
    gold["stop"] = int(g["stop"])
    compare_lightglue(out, gold)


def _full_inputs(seed, m, n):
    case = {"seed": seed, "m": m, "n": n, "input_dim": 256, "size0": (1024.0, 1024.0), "size1": (1024.0, 1024.0)}
    return gc.lg_inputs(case)


@pytest.mark.parametrize("mode", ["fixed", "default"])
def test_lightglue_gpu_full_size_vs_oracle(hip_lib, mode):
    """BASELINE config 3: 2048 x 2048 keypoints, 9 layers; fixed-work and reference-default modes."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0} if mode == "fixed" else \
           {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    f0, f1 = _full_inputs(21, 2048, 2048)
    net = _lg().LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
    out = _cpu(net(_data(f0, f1), dense=True))
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
    res = compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    print(mode, res, "stop", out["stop"], "S", out["matches"][0].shape[0])


@pytest.mark.parametrize("mode", ["fixed", "default"])
def test_lightglue_gpu_full_size_kv_images_from_the_projection_gemm(hip_lib, mode):
    """The large-batch path (128 x 256 GEMM blocks whose epilogue writes the K | V tile images: rotary + pre-split fused,
    cross-attention Q taken from the K image), forced at batch 1 with dim_tune_set(6, 2): same oracle, same tolerances; and
    a real large batch (8 ragged pairs, which selects that path by itself) equals the pairs run one by one on it."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0} if mode == "fixed" else \
           {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    f0, f1 = _full_inputs(21, 2048, 2048)
    lg = _lg()
    try:
        hip_lib.dim_tune_set(6, 2)
        hip_lib.dim_tune_set(11, 4)      # ... and the one-kernel feed-forward (the other large-batch-only kernel)
        net = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
        out = _cpu(net(_data(f0, f1), dense=True))
        ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
        res = compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
        print(mode, "fused", res, "stop", out["stop"], "S", out["matches"][0].shape[0])
        # 8 ragged pairs of the same images: the batch picks the path itself (dim_tune_set(6, 1)); singles are forced onto it
        g = torch.Generator().manual_seed(4)
        counts = [2048, 1900, 1777, 2048, 1500, 2001, 1999, 1024, 2048]
        kt = torch.rand(9, 2048, 2, generator=g) * 1024
        dt = torch.nn.functional.normalize(torch.randn(9, 2048, 256, generator=g), dim=-1)
        nt, st = torch.tensor(counts, dtype=torch.int32), torch.tensor([[1024.0, 1024.0]] * 9)
        pairs = torch.tensor([[i, i + 1] for i in range(8)], dtype=torch.int32)
        singles = []
        for a, b in pairs.tolist():
            data = {"image0": {"keypoints": kt[a, :counts[a]][None], "descriptors": dt[a, :counts[a]][None], "image_size": st[a][None]},
                    "image1": {"keypoints": kt[b, :counts[b]][None], "descriptors": dt[b, :counts[b]][None], "image_size": st[b][None]}}
            singles.append(_cpu(net(data)))
        hip_lib.dim_tune_set(6, 1)
        hip_lib.dim_tune_set(11, 3)
        big = lg.LightGlueHIP(sd, conf, max_pairs=8, max_kpts=2048)
        o = {k: v.cpu() for k, v in big.match_batch(kt.cuda(), dt.cuda(), nt.cuda(), st.cuda(), pair_idx=pairs.cuda()).items()}
        for p, r in enumerate(singles):
            S = int(o["n_matches"][p])
            assert int(o["stop"][p]) == r["stop"]
            assert torch.equal(o["matches"][p, :S], r["matches"][0])
            # (not bit-equal: a 2-item launch cuts the key range of every attention workgroup in 4 and merges the partial
            # softmaxes, a 16-item launch does not — fp32 reassociation)
            torch.testing.assert_close(o["scores"][p, :S], r["scores"][0], rtol=1e-3, atol=1e-9)
    finally:
        hip_lib.dim_tune_set(6, 1)
        hip_lib.dim_tune_set(11, 3)


def test_lightglue_gpu_batch_equals_single_and_pair_index(hip_lib):
    """A batch of ragged pairs through pair_idx == the same pairs one by one."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(5, 256, gain=2.0)
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    lg = _lg()
    g = torch.Generator().manual_seed(3)
    n_img, cap = 4, 300
    counts = [300, 0, 123, 257]  # includes an image with no keypoints
    kt = torch.rand(n_img, cap, 2, generator=g) * 640
    dt = torch.nn.functional.normalize(torch.randn(n_img, cap, 256, generator=g), dim=-1)
    nt = torch.tensor(counts, dtype=torch.int32)
    st = torch.tensor([[480.0, 640.0]] * n_img)
    pairs = torch.tensor([[0, 2], [2, 3], [0, 1], [3, 0]], dtype=torch.int32)
    net = lg.LightGlueHIP(sd, conf, max_pairs=4, max_kpts=cap)
    o = net.match_batch(kt.cuda(), dt.cuda(), nt.cuda(), st.cuda(), pair_idx=pairs.cuda())
    o = {k: v.cpu() for k, v in o.items()}
    single = lg.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap)
    for p, (a, b) in enumerate(pairs.tolist()):
        data = {"image0": {"keypoints": kt[a, :counts[a]][None], "descriptors": dt[a, :counts[a]][None], "image_size": st[a][None]},
                "image1": {"keypoints": kt[b, :counts[b]][None], "descriptors": dt[b, :counts[b]][None], "image_size": st[b][None]}}
        r = _cpu(single(data))
        S = int(o["n_matches"][p])
        assert int(o["stop"][p]) == r["stop"]
        assert torch.equal(o["matches"][p, :S], r["matches"][0])
        assert torch.equal(o["scores"][p, :S], r["scores"][0])
        assert torch.equal(o["matches01"][p, 0, :counts[a]].long(), r["matches0"][0])
        if counts[a] == 0 or counts[b] == 0:
            assert S == 0 and r["stop"] == 1


def test_ffn_layernorm_gelu_fused_op_at_production_rows(hip_lib):
    """ffn.0 -> LayerNorm -> GELU as one kernel (gemm_x6.hip, 64 x 512 blocks) vs an fp64 evaluation at a row count that puts
    TWO workgroups on every CU (65 536 rows = 1024 workgroups), twice.  The first version of its row reduction was correct on
    the emulator and at one workgroup per CU and returned ~1 row in 1000 with stale statistics here (and different ones on
    every run) — only this test, at this size, showed it."""
    from tests.test_ops_emu import _ffn_ln_gelu_case
    for M in (2048 + 37, 65536):
        C1, ref = _ffn_ln_gelu_case(hip_lib, M, 512, seed=M, device="cuda")
        C2, _ = _ffn_ln_gelu_case(hip_lib, M, 512, seed=M, device="cuda")
        err = (C1.double() - ref).abs().max(1).values
        assert int((err > 1e-5).sum()) == 0 and float(err.max()) < 1e-5, (M, float(err.max()), (err > 1e-5).nonzero().reshape(-1)[:8].tolist())
        assert torch.equal(C1, C2), M


def test_ffn_fused_op_at_production_rows(hip_lib):
    """The whole feed-forward as one kernel (gemm_x6_ffn_fused_kernel: hidden tile register-resident, ffn.3's K split over the four
    waves, partial sums exchanged through LDS) vs fp64 at row counts that put two workgroups on every CU, twice."""
    from tests.test_ops_emu import _ffn_fused_case
    for M in (2048 + 37, 65536):
        C1, ref = _ffn_fused_case(hip_lib, M, 512, seed=M, device="cuda")
        C2, _ = _ffn_fused_case(hip_lib, M, 512, seed=M, device="cuda")
        err = (C1.double() - ref).abs().max(1).values
        assert int((err > 2e-5).sum()) == 0, (M, float(err.max()), (err > 2e-5).nonzero().reshape(-1)[:8].tolist())
        assert torch.equal(C1, C2), M


def test_match_list_flip_rate_at_2048_within_the_measured_fp32_envelope(hip_lib):
    """BASELINE size (2048 x 2048 keypoints, fixed work): the first 8 pairs of the 200-pair study (scripts/study/lg_flip_rate.py;
    inputs with true correspondences, 124 .. 439 matches per pair), HIP vs the fp32 oracle at threshold 0 AND 0.1.  The bound is not
    a constant picked after a failure: it is the flip rate / decision margin the reference's own arithmetic shows against its fp64
    evaluation in that study (tests/test_configs_gpu.py::flip_rate_basis reads profiles/r04_flip_rate_summary.json)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "scripts" / "study"))
    import lg_flip_rate as study
    from tests.parity import match_list_difference_is_a_tie
    from tests.test_configs_gpu import _record, flip_rate_basis
    basis = flip_rate_basis()
    lg = importlib.import_module("deep-image-matching_amd.lightglue_hip")
    total = {0.0: 0, 0.1: 0}
    flips = {0.0: [], 0.1: []}
    mats = {}
    for p in range(8):
        c = study.case_of(p)
        sd, f = gc.lg_weights(c), gc.lg_inputs(c)
        if c["wseed"] not in mats:
            mats[c["wseed"]] = lg.LightGlueHIP(sd, study.CONF, max_pairs=1, max_kpts=2048)
        res = mats[c["wseed"]]({"image0": {"keypoints": f[0]["kpts"][None], "descriptors": f[0]["desc"][None], "image_size": f[0]["size"][None]},
                                "image1": {"keypoints": f[1]["kpts"][None], "descriptors": f[1]["desc"][None], "image_size": f[1]["size"][None]}})
        ref = lightglue_ref.lightglue_forward(f[0]["kpts"], f[0]["desc"], f[0]["size"], f[1]["kpts"], f[1]["desc"], f[1]["size"], sd, study.CONF, taps=True)
        mo, so = res["matches"][0].cpu(), res["scores"][0].cpu()
        for th in (0.0, 0.1):
            got, want = mo[so > th], ref["matches"][ref["scores"] > th]
            total[th] += int(want.shape[0])
            if not torch.equal(got, want):
                flips[th] += match_list_difference_is_a_tie(got, want, ref["log_assignment"], th, tie_tol=basis["tie_tol"])
    assert total[0.0] >= 8 * 100
    for th in (0.0, 0.1):
        assert len(flips[th]) <= 3 * basis["rate"] * total[th] + 1, (th, flips[th], basis)
    _record({"test": "lg_flip_rate_2048_first_8_pairs", "matches": total, "flips": flips, "basis": basis})


def test_one_pair_adaptive_depth_deferred_assignment_and_followed_stop_flags_on_hardware(hip_lib):
    """One pair per call with the reference's default confidences (the plugin hooks), pairs DESIGNED to stop after 3 / 5 / 7 / 9 layers
    (workloads.adaptive_lightglue_workload): round 6 evaluates the assignment once after the layer loop (dim_tune_set key 17) and lets the host follow
    the stop flags through mapped page-locked memory, two layers behind the device, to stop enqueueing layers (key 18) — real concurrency between the
    host's spin and lg_decide_kernel's system-scope stores, which the emulator cannot show.  Every setting must give the same stop layer and bit for bit
    the same matches, scores and dense log-assignment, repeatedly (the sequence number guards against a stale word of the previous call)."""
    wl = importlib.import_module("deep-image-matching_amd.workloads")
    stops = (3, 5, 7, 9)
    sd, kp, de, cnt, sz, expect = wl.adaptive_lightglue_workload(len(stops), n_kpts=1024, stops=stops)
    kp, de, cnt, sz = kp.cuda(), de.cuda(), cnt.cuda(), sz.cuda()
    conf = {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1}
    ref = {}
    try:
        for k17, k18 in ((0, 0), (1, 0), (1, 1)):
            assert hip_lib.dim_tune_set(17, k17) == 0 and hip_lib.dim_tune_set(18, k18) == 0
            net = _lg().LightGlueHIP(sd, conf, max_pairs=1, max_kpts=1024)
            for rep in range(3):
                for p, s in enumerate(stops):
                    pi = torch.tensor([[2 * p, 2 * p + 1]], dtype=torch.int32, device="cuda")
                    o = net.match_batch(kp, de, cnt, sz, pair_idx=pi, dense=True)
                    torch.cuda.synchronize()
                    S = int(o["n_matches"][0])
                    assert int(o["stop"][0]) == s == int(expect[p])
                    got = (o["matches"][0, :S].cpu(), o["scores"][0, :S].cpu(), o["dense"][0].cpu())
                    if p not in ref:
                        ref[p] = got
                        assert S > 100
                    else:
                        assert all(torch.equal(a, b) for a, b in zip(ref[p], got)), (k17, k18, rep, p)
    finally:
        hip_lib.dim_tune_set(17, 1)
        hip_lib.dim_tune_set(18, 1)


@pytest.mark.parametrize("kf16,df16,dn", [(0, 0, 0), (1, 1, 1), (0, 1, 0), (1, 0, 1)])
def test_lg_stage_features_on_hardware_equals_the_host_conversion(hip_lib, kf16, df16, dn):
    """dim_lg_stage_features at the headline size (2048 / 1777 keypoints, D = 256, table capacity 2048): float16 / float32 arrays in (N, D) / (D, N) layout ->
    the fp32 (N, D) feature table, bit for bit the host conversion (transpose + astype(float32)); rows past the live counts are zero."""
    import ctypes
    capi = importlib.import_module("deep-image-matching_amd.capi")
    g = np.random.default_rng(11 + kf16 + 2 * df16 + 4 * dn)
    D, cap, counts = 256, 2048, (2048, 1777)
    dev, descr, want = [], [], []
    for n in counts:
        k = (g.random((n, 2)) * 1000).astype(np.float16 if kf16 else np.float32)
        d = g.standard_normal((D, n) if dn else (n, D)).astype(np.float16 if df16 else np.float32)
        kt, dt = torch.from_numpy(k).cuda(), torch.from_numpy(d).cuda()
        dev += [kt, dt]
        descr.append(capi.LgRawFeatures(kt.data_ptr(), dt.data_ptr(), n, kf16, df16, dn))
        want.append((k.astype(np.float32), (d.T if dn else d).astype(np.float32)))
    ktab = torch.full((2, cap, 2), -7.0, device="cuda")
    dtab = torch.full((2, cap, D), -7.0, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(hip_lib, hip_lib.dim_lg_stage_features(ctypes.byref(descr[0]), ctypes.byref(descr[1]), cap, D, capi.ptr(ktab), capi.ptr(dtab), stream))
    torch.cuda.synchronize()
    for i, n in enumerate(counts):
        assert np.array_equal(ktab[i, :n].cpu().numpy(), want[i][0]) and np.array_equal(dtab[i, :n].cpu().numpy(), want[i][1])
        assert not ktab[i, n:].any() and not dtab[i, n:].any()


def test_key_split_attention_in_the_in_between_shapes_vs_oracle(hip_lib):
    """The key range of an attention launch is cut into the fewest parts (1, 2, 4) that give it two workgroups per CU (round 6: one pair of 2304 keypoints
    now takes 4 parts of 18 tiles instead of 2 of 36, three pairs of 2048 take 2 instead of 1).  A ragged one-pair call in the first shape against the
    oracle; a three-pair call in the second against the same pairs run alone (which the full-size test pins to the oracle), fixed work."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
    f0, f1 = _full_inputs(33, 2304, 2100)
    net = _lg().LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2304)
    out = _cpu(net(_data(f0, f1), dense=True))
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)
    compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    g = torch.Generator().manual_seed(6)
    counts = [2048, 1900, 1777, 2048, 2001, 1500]
    kt = torch.rand(6, 2048, 2, generator=g) * 1024
    dt = torch.nn.functional.normalize(torch.randn(6, 2048, 256, generator=g), dim=-1)
    nt, st = torch.tensor(counts, dtype=torch.int32), torch.tensor([[1024.0, 1024.0]] * 6)
    big = _lg().LightGlueHIP(sd, conf, max_pairs=3, max_kpts=2048)
    o = {k: v.cpu() for k, v in big.match_batch(kt.cuda(), dt.cuda(), nt.cuda(), st.cuda()).items()}
    one = _lg().LightGlueHIP(sd, conf, max_pairs=1, max_kpts=2048)
    for p in range(3):
        a, b = 2 * p, 2 * p + 1
        data = {"image0": {"keypoints": kt[a, :counts[a]][None], "descriptors": dt[a, :counts[a]][None], "image_size": st[a][None]},
                "image1": {"keypoints": kt[b, :counts[b]][None], "descriptors": dt[b, :counts[b]][None], "image_size": st[b][None]}}
        r = _cpu(one(data))
        S = int(o["n_matches"][p])
        assert int(o["stop"][p]) == r["stop"] and torch.equal(o["matches"][p, :S], r["matches"][0])
        torch.testing.assert_close(o["scores"][p, :S], r["scores"][0], rtol=1e-3, atol=1e-9)
