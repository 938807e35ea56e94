"""CPU: the HIP LightGlue sources, compiled against the test-only emulator, vs the oracle and the
reference's golden vectors (early stop, pruning, the empty exit, 128-d input_proj)."""
import importlib
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lightglue_ref
from tests import golden_cases as gc
from tests.parity import compare_lightglue

lg_mod = importlib.import_module("deep-image-matching_amd.lightglue_hip")
GOLD = Path(__file__).parent / "golden"


_REF_CACHE = {}


def run_case(lib, case, device="cpu"):
    sd = gc.lg_weights(case)
    f0, f1 = gc.lg_inputs(case)
    net = lg_mod.LightGlueHIP(sd, case["conf"], max_pairs=1, max_kpts=max(case["m"], case["n"]), device=device, lib=lib)
    data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
    out = net(data, dense=True)
    out = {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in out.items()}
    key = repr(sorted(case.items(), key=lambda kv: kv[0]))
    if key not in _REF_CACHE:   # (the oracle's output for a case does not depend on the library under test: several tests run the same case with other kernel selections)
        _REF_CACHE[key] = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, case["conf"], taps=True)
    return out, _REF_CACHE[key]


@pytest.mark.parametrize("name", list(gc.LG_CASES))
def test_lightglue_emulated_vs_golden_and_oracle(emu_lib, name):
    case = gc.LG_CASES[name]
    out, ref = run_case(emu_lib, case)
    res = compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    g = np.load(GOLD / f"lg_{name}.npz")
    gold = {k: torch.from_numpy(np.asarray(g[k])) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1",
                                                             "matches", "scores", "prune0", "prune1")}
    gold["stop"] = int(g["stop"])
    compare_lightglue(out, gold)


def test_lightglue_bf16x6_mode_still_matches_the_oracle(emu_lib):
    """The default split mode is fp16x3 (two fp16 planes, three MFMA terms); the bf16x6 mode (three bf16
    planes, six terms) and the plain fp32-MFMA mode stay selectable (dim_tune_set key 1) and parity-green."""
    name = next(iter(gc.LG_CASES))
    try:
        for mode in (1, 0):
            emu_lib.dim_tune_set(1, mode)
            out, ref = run_case(emu_lib, gc.LG_CASES[name])
            compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    finally:
        emu_lib.dim_tune_set(1, 2)


@pytest.mark.parametrize("name", list(gc.LG_CASES))
def test_lightglue_kv_images_written_by_the_projection_gemm(emu_lib, name):
    """Large batches run the 128 x 256 GEMM block, whose epilogue writes the attention kernel's K | V tile images itself
    (rotary + pre-split; cross attention then takes Q from the item's own K image).  dim_tune_set(6, 2) forces that path at
    the golden sizes (ragged key counts, pruning, early stop, 128-d inputs): same goldens, same oracle."""
    case = gc.LG_CASES[name]
    try:
        emu_lib.dim_tune_set(6, 2)
        out, ref = run_case(emu_lib, case)
    finally:
        emu_lib.dim_tune_set(6, 1)
    compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    g = np.load(GOLD / f"lg_{name}.npz")
    gold = {k: torch.from_numpy(np.asarray(g[k])) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1",
                                                             "matches", "scores", "prune0", "prune1")}
    gold["stop"] = int(g["stop"])
    compare_lightglue(out, gold)


def test_attention_tile_dma_is_ordered_both_ways(emu_lib):
    """As test_lds_dma_staging_is_ordered_both_ways (test_superpoint_emu.py), for the double-buffered key-tile image of the
    attention kernel: transfers landing at the wait (default) and at issue must give the same result."""
    name = next(iter(gc.LG_CASES))
    out, _ = run_case(emu_lib, gc.LG_CASES[name])
    try:
        emu_lib.hipemu_set_dma_mode(1)
        emu_lib.hipemu_set_schedule(1)   # wave after wave between barriers: the waves are maximally out of step
        out2, _ = run_case(emu_lib, gc.LG_CASES[name])
    finally:
        emu_lib.hipemu_set_dma_mode(0)
        emu_lib.hipemu_set_schedule(1 if os.environ.get("HIPEMU_ORDER") == "wave_serial" else 0)
    assert torch.equal(out["dense"], out2["dense"]) and torch.equal(out["matches"][0], out2["matches"][0])


def test_key_split_attention_with_two_tiles_of_prefetch_distance(emu_research_lib):
    """One pair per call (the plugin hooks): the attention launches cut the key range into 4 parts.  Round-5 prototype (research build, knob 12 = 22;
    measured slower on hardware): TWO key-tile images in flight (three LDS buffers, `s_waitcnt vmcnt(4)` = retire the oldest
    transfer only, a plain s_barrier).  The golden cases have one tile per part — this one has 4 / 3 (ragged last part), so the ring wraps: equal bit
    for bit to the product's one-tile-ahead kernel, with the emulated transfers landing at the wait (a wait that retires too little reads a stale tile
    — the emulator's s_waitcnt retires all but the N newest transfers of a thread) AND at issue with the waves maximally out of step (a transfer into a
    buffer that is still being read corrupts it); the product kernel equals the oracle."""
    lib = emu_research_lib
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(7, 256, n_layers=1, gain=2.0)
    conf = {"n_layers": 1, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0}
    g = torch.Generator().manual_seed(11)
    n0, n1 = 500, 361
    f = [{"kpts": torch.rand(n, 2, generator=g) * 640, "desc": torch.nn.functional.normalize(torch.randn(n, 256, generator=g), dim=-1)} for n in (n0, n1)]
    size = torch.tensor([480.0, 640.0])
    data = {"image0": {"keypoints": f[0]["kpts"][None], "descriptors": f[0]["desc"][None], "image_size": size[None]},
            "image1": {"keypoints": f[1]["kpts"][None], "descriptors": f[1]["desc"][None], "image_size": size[None]}}
    outs = []
    try:
        for knob, dma_early in ((0, 0), (22, 0), (22, 1)):
            assert lib.dim_tune_set(12, knob) == 0
            lib.hipemu_set_dma_mode(dma_early)
            lib.hipemu_set_schedule(1 if dma_early else 0)
            net = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=n0, device="cpu", lib=lib)
            outs.append(net(data, dense=True))
    finally:
        lib.dim_tune_set(12, 0)
        lib.hipemu_set_dma_mode(0)
        lib.hipemu_set_schedule(1 if os.environ.get("HIPEMU_ORDER") == "wave_serial" else 0)
    for o in outs[1:]:
        assert torch.equal(outs[0]["dense"], o["dense"]) and torch.equal(outs[0]["matches"][0], o["matches"][0])
    ref = lightglue_ref.lightglue_forward(f[0]["kpts"], f[0]["desc"], size, f[1]["kpts"], f[1]["desc"], size, sd, conf, taps=True)
    compare_lightglue(outs[0], ref, dense_ref=ref.get("log_assignment"), dense_out=outs[0]["dense"])


def test_batch_with_ragged_and_empty_images_on_the_projection_written_images(emu_lib):
    """A batch of ragged pairs (one image without keypoints) through pair_idx on the large-batch path (K | V tile images written
    by the projection GEMM, forced with dim_tune_set(6, 2)), adaptive depth and width on: every pair equals the same pair
    run alone."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(5, 256, n_layers=3, gain=2.0)
    conf = {"n_layers": 3, "depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.0}
    g = torch.Generator().manual_seed(3)
    n_img, cap = 4, 70
    counts = [70, 0, 33, 57]
    kt = torch.rand(n_img, cap, 2, generator=g) * 640
    dt = torch.nn.functional.normalize(torch.randn(n_img, cap, 256, generator=g), dim=-1)
    nt = torch.tensor(counts, dtype=torch.int32)
    st = torch.tensor([[480.0, 640.0]] * n_img)
    pairs = torch.tensor([[0, 2], [2, 3], [0, 1], [3, 0]], dtype=torch.int32)
    try:
        emu_lib.dim_tune_set(6, 2)
        capi = importlib.import_module("deep-image-matching_amd.capi")
        net = lg_mod.LightGlueHIP(sd, conf, max_pairs=4, max_kpts=cap, device="cpu", lib=emu_lib)
        capi.saturation(emu_lib, None, reset=True)
        o = net.match_batch(kt, dt, nt, st, pair_idx=pairs)
        # benign inputs: the fp16x3 range guard must stay silent — in particular for the rows past each item's ragged end
        # (the emulator hands out device memory poisoned with NaNs: a table row nothing initialised would trip it)
        total, sites = capi.saturation(emu_lib, None, reset=True)
        assert total == 0, sites
        single = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap, device="cpu", lib=emu_lib)
        for p, (a, b) in enumerate(pairs.tolist()):
            data = {"image0": {"keypoints": kt[a, :counts[a]][None], "descriptors": dt[a, :counts[a]][None], "image_size": st[a][None]},
                    "image1": {"keypoints": kt[b, :counts[b]][None], "descriptors": dt[b, :counts[b]][None], "image_size": st[b][None]}}
            r = single(data)
            S = int(o["n_matches"][p])
            assert int(o["stop"][p]) == r["stop"]
            assert torch.equal(o["matches"][p, :S], r["matches"][0]) and torch.equal(o["scores"][p, :S], r["scores"][0])
            assert torch.equal(o["matches01"][p, 0, :counts[a]].long(), r["matches0"][0])
            if counts[a] == 0 or counts[b] == 0:
                assert S == 0 and r["stop"] == 1
    finally:
        emu_lib.dim_tune_set(6, 1)


def test_deferred_assignment_equals_the_gated_launches_after_every_layer(emu_lib):
    """Adaptive depth: a pair that stops at layer i is skipped by every later launch, so its assignment can wait.  dim_tune_set(17, 1) (default) runs the
    final projection / similarity / double softmax / arg-max ONCE after the layer loop, every item reading the weights of its own stop layer from per-layer
    tables; (17, 0) launches them, gated on the stop layer, after every layer.  A ragged batch whose pairs stop at different layers (incl. the
    no-keypoints exit and pruning): every output bit for bit equal, dense log-assignment included; and the stop layers really differ."""
    weights = importlib.import_module("deep-image-matching_amd.weights")
    sd = weights.synthetic_lightglue_state_dict(5, 256, n_layers=4, gain=2.0)
    sd = {k: v.clone() for k, v in sd.items()}
    for i in range(3):   # token-confidence biases that make the layers differently sure of themselves: pairs leave at different depths
        sd[f"token_confidence.{i}.token.0.bias"] = sd[f"token_confidence.{i}.token.0.bias"] + (-2.0, 1.5, 3.0)[i]
    conf = {"n_layers": 4, "depth_confidence": 0.6, "width_confidence": 0.99, "filter_threshold": 0.0, "pruning_min_kpts": -1}
    g = torch.Generator().manual_seed(4)
    n_img, cap = 5, 90
    counts = [90, 0, 41, 77, 64]
    kt = torch.rand(n_img, cap, 2, generator=g) * 640
    dt = torch.nn.functional.normalize(torch.randn(n_img, cap, 256, generator=g), dim=-1)
    dt[4] = dt[3] + 0.05 * torch.randn(cap, 256, generator=g)     # a pair with real correspondences behaves differently from noise pairs
    nt = torch.tensor(counts, dtype=torch.int32)
    st = torch.tensor([[480.0, 640.0]] * n_img)
    pairs = torch.tensor([[0, 2], [3, 4], [0, 1], [4, 3], [2, 3]], dtype=torch.int32)
    outs = []
    try:
        for defer in (1, 0):
            assert emu_lib.dim_tune_set(17, defer) == 0
            net = lg_mod.LightGlueHIP(sd, conf, max_pairs=5, max_kpts=cap, device="cpu", lib=emu_lib)
            outs.append(net.match_batch(kt, dt, nt, st, pair_idx=pairs, dense=True))
    finally:
        emu_lib.dim_tune_set(17, 1)
    a, b = outs
    for k in ("n_matches", "stop", "dense"):
        assert torch.equal(a[k], b[k]), k
    for p, (i0, i1) in enumerate(pairs.tolist()):   # (the tables are only defined up to the live counts)
        S = int(a["n_matches"][p])
        assert torch.equal(a["matches"][p, :S], b["matches"][p, :S]) and torch.equal(a["scores"][p, :S], b["scores"][p, :S])
        for side, img in enumerate((i0, i1)):
            n = counts[img]
            for k in ("matches01", "mscores01", "prune01"):
                assert torch.equal(a[k][p, side, :n], b[k][p, side, :n]), (k, p, side)
    assert len(set(a["stop"].tolist())) >= 2, a["stop"]
    assert int(a["n_matches"].sum()) > 0
    # one pair per call (the plugin hooks): the host follows the stop flags two layers behind and stops enqueueing layers (dim_tune_set key 18) — every pair
    # alone, followed and not followed: bit for bit the same, and the same matches as its row of the batch (whose launches take other block shapes:
    # scores to fp32 rounding)
    assert min(a["stop"].tolist()) <= 2, a["stop"]     # (a pair that leaves early enough for the loop to be cut short)
    single = {}
    try:
        for follow in (1, 0):
            assert emu_lib.dim_tune_set(18, follow) == 0
            net1 = lg_mod.LightGlueHIP(sd, conf, max_pairs=1, max_kpts=cap, device="cpu", lib=emu_lib)
            for p in range(len(pairs)):
                o = net1.match_batch(kt, dt, nt, st, pair_idx=pairs[p:p + 1].contiguous(), dense=True)
                S = int(a["n_matches"][p])
                assert int(o["n_matches"][0]) == S and int(o["stop"][0]) == int(a["stop"][p])
                assert torch.equal(o["matches"][0, :S], a["matches"][p, :S]) and (o["scores"][0, :S] - a["scores"][p, :S]).abs().max().item() < 1e-5 if S else True
                single[(follow, p)] = (o["matches"][0, :S].clone(), o["scores"][0, :S].clone(), o["dense"][0].clone())
    finally:
        emu_lib.dim_tune_set(18, 1)
    for p in range(len(pairs)):
        assert all(torch.equal(x, y) for x, y in zip(single[(1, p)], single[(0, p)])), p


def test_ffn_layernorm_gelu_epilogue_equals_the_separate_pass(emu_lib):
    """dim_tune_set(11, 2) forces the 64 x 512 ffn.0 block whose epilogue applies LayerNorm(512) + erf-GELU (the production
    path at large batches) at the golden sizes; (11, 0) keeps ffn.0 -> lg_ln_gelu_kernel.  Same statistics (two-pass mean /
    centred variance), different reduction tree: the scores agree to fp32 rounding, every integer output is identical, and
    both equal the oracle and the reference goldens (ragged counts, adaptive depth and width, 128-d inputs)."""
    n_diff = 0
    for name, case in list(gc.LG_CASES.items())[:2]:     # ragged counts, adaptive depth, fixed work; the other cases add nothing to these kernels
        outs = {}
        try:
            for mode in (0, 2, 4):
                emu_lib.dim_tune_set(11, mode)
                out, ref = run_case(emu_lib, case)
                compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
                outs[mode] = out
        finally:
            emu_lib.dim_tune_set(11, 3)
        for other in (2, 4):      # 4 = the whole feed-forward in one kernel (hidden tile register-resident, ffn.3's K split over the waves)
            a, b = outs[0], outs[other]
            assert torch.equal(a["matches0"], b["matches0"]) and torch.equal(a["matches"][0], b["matches"][0]) and int(a["stop"]) == int(b["stop"]), (name, other)
            if a["matching_scores0"].numel():
                assert (a["matching_scores0"] - b["matching_scores0"]).abs().max().item() < 2e-5, (name, other)
                n_diff += int(not torch.equal(a["dense"], b["dense"]))
    assert n_diff > 0        # two different code paths really ran


def test_assignment_fast_and_generic_kernels_agree(emu_lib):
    """The assignment passes exist twice: 16-byte single-read kernels for tables of up to 2048 keypoints (the row stride is always a
    multiple of 4), 4-byte generic ones beyond.  The same ragged pair through both (max_kpts 96 vs 2052): identical integer outputs,
    scores and the dense log-assignment equal to fp32 rounding — and both equal to the oracle."""
    case = gc.LG_CASES["default"]
    sd = gc.lg_weights(case)
    f0, f1 = gc.lg_inputs(case)
    data = {"image0": {"keypoints": f0["kpts"][None], "descriptors": f0["desc"][None], "image_size": f0["size"][None]},
            "image1": {"keypoints": f1["kpts"][None], "descriptors": f1["desc"][None], "image_size": f1["size"][None]}}
    ref = lightglue_ref.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, case["conf"], taps=True)
    m, n = case["m"], case["n"]
    net = lg_mod.LightGlueHIP(sd, case["conf"], max_pairs=1, max_kpts=96, device="cpu", lib=emu_lib)
    a = net(data, dense=True)
    a = {k: ([t.cpu() for t in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in a.items()}
    compare_lightglue(a, ref, dense_ref=ref["log_assignment"], dense_out=a["dense"])
    # launch shapes and kernel selections follow the feature table's rows per image (round 6: not the handle's capacity): the same pair in a 2052-row table
    # takes the 16-byte kernels with 16 chunks per lane (rows of up to 4096 live columns), in a 4100-row table the generic ones
    prev = None
    for cap in (2052, 4100):
        big = lg_mod.LightGlueHIP(sd, case["conf"], max_pairs=1, max_kpts=cap, device="cpu", lib=emu_lib)
        kt, dt = torch.zeros(2, cap, 2), torch.zeros(2, cap, f0["desc"].shape[1])
        kt[0, :m], kt[1, :n], dt[0, :m], dt[1, :n] = f0["kpts"], f1["kpts"], f0["desc"], f1["desc"]
        b = big.match_batch(kt, dt, torch.tensor([m, n], dtype=torch.int32), torch.stack([f0["size"], f1["size"]]), dense=True)
        S = int(b["n_matches"][0])
        assert int(b["stop"][0]) == a["stop"] and torch.equal(b["matches"][0, :S], a["matches"][0])
        assert torch.equal(b["matches01"][0, 0, :m].long(), a["matches0"].reshape(-1).long()) and torch.equal(b["matches01"][0, 1, :n].long(), a["matches1"].reshape(-1).long())
        # (the larger table also changes launch shapes upstream — attention key splits, GEMM blocks: fp32 noise of the whole network)
        assert (a["dense"][:m, :n] - b["dense"][0, :m, :n]).abs().max().item() < 5e-4
        if cap > 4096:
            assert not torch.equal(a["dense"][:m, :n], b["dense"][0, :m, :n])      # two different code paths really ran (the 16-chunk kernels reduce in the 8-chunk kernels' order: equal bits)
        if prev is not None:
            assert (prev - b["dense"][0, :m, :n]).abs().max().item() < 5e-4
        prev = b["dense"][0, :m, :n].clone()
        del big


def test_swapping_the_images_transposes_the_result(emu_lib):
    """A size-independent property of the network (LGN:146-211: the same weights serve both images, the cross block is symmetric):
    match(A, B) and match(B, A) are transposes of each other — matches0 <-> matches1, stop layer equal, scores to fp32 rounding."""
    case = gc.LG_CASES["default"]
    sd = gc.lg_weights(case)
    f0, f1 = gc.lg_inputs(case)
    net = lg_mod.LightGlueHIP(sd, case["conf"], max_pairs=1, max_kpts=max(case["m"], case["n"]), device="cpu", lib=emu_lib)
    side = lambda f: {"keypoints": f["kpts"][None], "descriptors": f["desc"][None], "image_size": f["size"][None]}
    ab = net({"image0": side(f0), "image1": side(f1)})
    ba = net({"image0": side(f1), "image1": side(f0)})
    assert int(ab["stop"]) == int(ba["stop"])
    assert torch.equal(ab["matches0"].cpu(), ba["matches1"].cpu()) and torch.equal(ab["matches1"].cpu(), ba["matches0"].cpu())
    assert (ab["matching_scores0"].cpu() - ba["matching_scores1"].cpu()).abs().max().item() < 1e-5
    assert torch.equal(ab["prune0"].cpu(), ba["prune1"].cpu()) and torch.equal(ab["prune1"].cpu(), ba["prune0"].cpu())
    ma, mb = ab["matches"][0].cpu(), ba["matches"][0].cpu()
    assert ma.shape[0] > 0 and {tuple(x) for x in ma.tolist()} == {(j, i) for i, j in mb.tolist()}


def test_wide_gemm_blocks_with_64_wide_k_chunks_are_bit_identical(emu_research_lib):
    """RESEARCH build (-DDIM_RESEARCH: the product library does not contain these prototypes).  dim_tune_set(14, 64): the pipelined 128 x 256 GEMM block and the q|k|v kernel stage 64 instead of 32 K values per barrier pair;
    dim_tune_set(14, 33): the staged activation tile is double-buffered in LDS, one barrier per chunk; dim_tune_set(14, 37): their fragment sets refilled tile by tile; dim_tune_set(14, 36): the fused
    feed-forward's previous K loop (the product one re-requests a column tile's weight fragments right after its MFMAs).  Same
    k-step order, same MFMA sequence per accumulator: every output bit for bit, with the large-batch kernels forced (6 = 2) on two
    golden cases."""
    for name in ("default", "fixed"):
        case = gc.LG_CASES[name]
        outs = []
        emu_lib = emu_research_lib
        try:
            emu_lib.dim_tune_set(6, 2)
            for kc in ((32, 64, 33, 36, 37, 256) if name == "fixed" else (32, 256)):   # 256 (round 5): 256-row blocks at the 512-register point   # the two prototypes on one case, the A/B loop of the product kernel on both
                assert emu_lib.dim_tune_set(14, kc) == 0
                out, ref = run_case(emu_lib, case)
                compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
                outs.append(out)
        finally:
            emu_lib.dim_tune_set(6, 1); emu_lib.dim_tune_set(14, 32)
        a = outs[0]
        for b in outs[1:]:
            assert torch.equal(a["dense"], b["dense"]) and torch.equal(a["matches0"], b["matches0"]) and torch.equal(a["matching_scores0"], b["matching_scores0"]), name


def test_lightglue_64_wide_descriptors_vs_oracle(emu_lib):
    """input_dim 64 (aliked-t16's descriptors through input_proj, LGN:361-364 with a custom `input_dim`): no golden — the reference's feature
    table has no 64-d entry — but the oracle (pinned on the 128- and 256-d goldens) runs the same code with any input_dim."""
    case = {**gc.LG_CASES["aliked_dim"], "input_dim": 64, "wseed": 7}
    out, ref = run_case(emu_lib, case)
    compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
    assert out["matches0"].shape[-1] == case["m"]


def test_fused_feed_forward_variants_are_bit_identical(emu_research_lib):
    """RESEARCH build, the one-kernel feed-forward forced on (11 = 4): the product loop (14 = 32), round 3's loop (36) and round 5's 128-row block at the
    512-register point (128: 16 ffn.0 accumulators per wave, the reduce-scatter of ffn.3 in two passes of four tiles) — the same terms in the same
    order for every output element: every bit equal, on a ragged adaptive case and the fixed-work case."""
    lib = emu_research_lib
    for name in ("default",):       # ragged counts, adaptive depth (the fixed-work case adds nothing to these kernels)
        case = gc.LG_CASES[name]
        outs = []
        try:
            lib.dim_tune_set(6, 2); lib.dim_tune_set(11, 4)
            for kc in (32, 36, 128):
                assert lib.dim_tune_set(14, kc) == 0
                out, ref = run_case(lib, case)
                compare_lightglue(out, ref, dense_ref=ref.get("log_assignment"), dense_out=out["dense"])
                outs.append(out)
        finally:
            lib.dim_tune_set(6, 1); lib.dim_tune_set(11, 3); lib.dim_tune_set(14, 32)
        for b in outs[1:]:
            assert torch.equal(outs[0]["dense"], b["dense"]) and torch.equal(outs[0]["matches0"], b["matches0"]), name
