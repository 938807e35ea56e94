"""Headline benchmark: image-pairs/s, SuperPoint+LightGlue, synthetic 1024x1024 pairs @2048 kpts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--pairs` synthetic pairs that are
already resident in HBM: SuperPoint extraction of both images of every pair (2*pairs images,
1024x1024, nms 3 / thr 0.0005 / top-2048) followed by LightGlue matching of every pair in
fixed-work mode (9 layers, 2048 x 2048, early stop and pruning off — BASELINE config 3's
roofline denominator).  Nothing is cached between steps and nothing is skipped.  Pairs are
sharded across ranks (weak scaling: every rank runs K steps of `--pairs` pairs) with no
data-path collective; after the last step the per-rank match tables are exchanged with ONE
RCCL all-gather (inside the timed region), as north_star asks.

Arithmetic: fp32 results ("dtype": "f32").  Matrix products run on the 16-bit matrix cores with every
fp32 operand (scaled by an exact power of two) split into two fp16 pieces and the three leading cross
terms accumulated in fp32 ("fp16x3", csrc/dim_common.h SplitMma<2>): fp32-class accuracy (hardware probe
scripts/probe/mfma_f16_probe.hip: 0.7-1.7e-7 of sum|a*b| where an fp32 fmaf chain gives 1.2-4.5e-7) at
3 fp16 MFMAs per product step instead of 8 fp32 MFMAs.  `dim_tune_set(1, 1)` selects the exact 3-way
bf16 split with six terms ("bf16x6"), `dim_tune_set(1, 0)` plain fp32 MFMA.

Default workload = BASELINE configs[2]: 50 pairs per step, so the 20 steps the driver asks for cover the 1000 pairs
that config names (a timed region of ~2 s: sustained clocks, not a burst).  `python bench.py --gpus N` without a
torchrun wrapper re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL).

Rank 0 prints one JSON line carrying the contract fields plus
  roofline     — the dominant kernel (conv3x3_x6_kernel<64,1,1,true,2>: conv1a fused into conv1b, 44 % of
                 SuperPoint's FLOPs) timed live with HIP events on the launch stream over the timed
                 region; `achieved` counts ALGORITHMIC fp32 FLOPs; `peak` is the dense fp16 MFMA peak of
                 MI355X_MICROARCH.md (2500 TFLOP/s: the precision actually issued, SURVEY §8(d) — the three
                 split-precision passes are NOT credited), so `frac` <= 1/3 by construction;
                 `mfma_pipe_util` = achieved x 3 passes / 2500 is the matrix-pipe utilisation the PMC counters
                 show (profiles/*_pmc_mfma_summary.txt) and `frac_of_fp32_mfma_peak` compares with plain fp32 MFMA;
  sustained_clock_mhz — average shader clock over the timed region (s_memtime / s_memrealtime probes);
  cpu_baseline — the reference modules (kind "reference", when /root/reference exists) or the oracle (kind
                 "port", on the GPU box) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PKG = "deep-image-matching_amd"

SP_GFLOP_PER_IMAGE = 177.85       # SURVEY.md §8(d), torch flop counter on the reference module
LG_GFLOP_PER_PAIR = 229.8         # 2048 x 2048, 9 layers, fixed work
CONV1A_GFLOP_PER_IMAGE = 2 * 0.604   # Appendix B: 3x3, 1->64 @1024^2 (+ReLU), fused into the conv1b kernel
CONV1B_GFLOP_PER_IMAGE = 2 * 38.655  # Appendix B: 3x3, 64->64 @1024^2 (+ReLU+pool)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
# What the matrix cores of an MI355X SUSTAIN on non-zero operands under the package power limit (scripts/probe/mfma_power_probe.hip,
# profiles/r04_mfma_power_probe.jsonl: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD, operands distributed like the fp16x3 kernels' —
# pieces of post-ReLU activations and of scaled weights; 1314 W, 1.65 GHz; 2450 with all-zero operands, 1580 with one ds_read_b128 per MFMA)
SUSTAINED_FP16_MFMA_TFLOPS = 1727.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
X6_PASSES = 3                     # fp16 MFMA terms per fp32-accurate product step (lh, hl, hh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (configs2: 20 x 50 pairs) / 1 (config4: one pass over the 10 000-pair job)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (configs2) / 1 (config4: one reduced pass)")
    ap.add_argument("--workload", choices=("configs2", "config4", "config5", "config1"), default="configs2", help="config1: BASELINE configs[0], the five sacre-coeur "
                    "photographs (tests/assets/config1) -> 10 brute-force pairs with config/superpoint+lightglue.yaml's parameters, through the per-call plugin hooks and "
                    "through BatchedImageMatcher, beside the reference's CPU path on the same inputs; config5: BASELINE configs[4], ALIKED + LightGlue on tiled "
                    "6000x4000 images through pipeline.TiledPairPipeline (strong scaling: `--images` images -> exhaustive pairs shared by the ranks); "
                    "configs2 (default, the headline line): BASELINE configs[2]; "
                    "config4: BASELINE configs[3], 150 images -> first 10 000 exhaustive pairs through PairMatchingPipeline "
                    "(phases 1-4 of SURVEY 8(e), STRONG scaling: the job is fixed, ranks share it)")
    ap.add_argument("--images", type=int, default=None, help="config4: 150 images; config5: 8 images of 6000 x 4000")
    ap.add_argument("--job-pairs", type=int, default=10000, help="config4 only")
    ap.add_argument("--pairs", type=int, default=50, help="pairs per step per GPU (50 x 20 steps = the 1000 pairs of configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="developer knob: path of an alternative libdim_hip build to load")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="developer knob: dim_tune_set(KEY, VALUE) before the "
                    "networks are created (A/B runs of kernel variants; the default line uses none)")
    ap.add_argument("--overlap", action="store_true", help="time the two-stream schedule (extraction of batch i+1 overlapping matching "
                    "of batch i) as the main region")
    ap.add_argument("--other-schedule", action="store_true", help="also time the K steps under the other schedule right after the main region "
                    "(reported as two_stream_overlap / single_stream); off by default: BENCH_r05 measured 605.22 vs 605.13 pairs/s")
    ap.add_argument("--no-overlap", action="store_true", help="(default since round 1) kept for compatibility")
    ap.add_argument("--main-region-only", action="store_true", help="skip the second (other-schedule) region: used for the rocprofv3 passes")
    ap.add_argument("--cpu-sample-pairs", type=int, default=5)
    ap.add_argument("--no-adaptive", action="store_true", help="skip the adaptive-depth / width sub-record (LightGlue's reference defaults on a batch that really adapts)")
    ap.add_argument("--tile-pair-batch", type=int, default=16, help="config5: tile pairs per dim_lg_match call")
    ap.add_argument("--tile-selection", default="PRESELECTION", help="config5: tile_selection method (PRESELECTION | GRID | EXHAUSTIVE | PRESELECTION_AFFINE_TRANSFORM)")
    ap.add_argument("--strong-timeout", type=float, default=300.0, help="seconds after which rank 0 prints the headline line without the strong_scaling sub-record and exits 3")
    ap.add_argument("--cpu-only", action="store_true", help="config1: run only the CPU leg (the reference modules when /root/reference exists): this is how the "
                    "build container produces profiles/*_config1_cpu_reference.json")
    ap.add_argument("--no-hook-path", action="store_true", help="skip the hook_path sub-record (batch-1 calls through the plugin classes)")
    ap.add_argument("--no-live-traffic", action="store_true", help="keep roofline.traffic at the value cited from profiles/conv1b_hbm_bytes.json instead of measuring it with two "
                    "rocprofv3 --pmc passes of a one-step child run (N = 1 only; skipped when rocprofv3 is not on PATH)")
    ap.add_argument("--no-strong-scaling", action="store_true", help="skip the strong_scaling sub-record (the config-4 job run after the timed region)")
    a = ap.parse_args()
    if a.images is None:
        a.images = 8 if a.workload == "config5" else 150
    if a.steps is None:
        a.steps = 20 if a.workload == "configs2" else (5 if a.workload == "config1" else 1)
    if a.warmup is None:
        a.warmup = 3 if a.workload == "configs2" else 1
    return a


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def _reference_modules():
    """The reference's own network files (SPN, LGN: torch/numpy only), imported by path when /root/reference exists
    (the build container); None on the GPU box."""
    ref = Path("/root/reference/src/deep_image_matching/thirdparty")
    spn, lgn = ref / "SuperGluePretrainedNetwork/models/superpoint.py", ref / "LightGlue/lightglue/lightglue.py"
    if not (spn.exists() and lgn.exists()):
        return None
    import importlib.util

    mods = []
    for path, name in ((spn, "ref_spn_bench"), (lgn, "ref_lgn_bench")):
        spec = importlib.util.spec_from_file_location(name, str(path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return mods


def cpu_baseline(n_pairs: int):
    """The reference modules (kind 'reference') when /root/reference is present, else the oracle (kind 'port'), on the host
    cores: same synthetic inputs, same configuration (SURVEY §8(d) "CPU reference timing")."""
    from oracle import lightglue_ref, superpoint_ref

    weights = importlib.import_module(PKG + ".weights")
    ref_mods = _reference_modules()
    # torch CPU peaks at 16-32 threads on this path (measured on the GPU box's 256-CPU host: 8 thr 0.25, 16 thr 0.28,
    # 32 thr 0.28, 64 thr 0.19, 128 thr 0.08 pairs/s), so the baseline runs at its best setting
    cores = min(16, os.cpu_count() or 16)
    torch.set_num_threads(cores)
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    lg_sd = weights.synthetic_lightglue_state_dict(0, 256)
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
    size = torch.tensor([1024.0, 1024.0])
    if ref_mods is not None:
        spn, lgn = ref_mods
        orig = torch.hub.load_state_dict_from_url
        torch.hub.load_state_dict_from_url = lambda *a_, **k_: sp_sd  # SPN:149 downloads; feed the same synthetic weights
        try:
            ref_sp = spn.SuperPoint(dict(cfg)).eval()
        finally:
            torch.hub.load_state_dict_from_url = orig
        ref_lg = lgn.LightGlue(features=None, input_dim=256, **conf).eval()
        ref_lg.load_state_dict(lg_sd, strict=False)

    @torch.no_grad()
    def one_pair(seed):
        f = []
        for s in (2 * seed, 2 * seed + 1):
            img = torch.rand(1, 1, 1024, 1024, generator=torch.Generator().manual_seed(s))
            if ref_mods is not None:
                o = ref_sp({"image": img})
                f.append({"keypoints": o["keypoints"][0], "descriptors": o["descriptors"][0]})
            else:
                f.append(superpoint_ref.superpoint_forward(img, sp_sd, cfg))
        if ref_mods is not None:
            ref_lg({"image0": {"keypoints": f[0]["keypoints"][None], "descriptors": f[0]["descriptors"].t()[None], "image_size": size[None]},
                    "image1": {"keypoints": f[1]["keypoints"][None], "descriptors": f[1]["descriptors"].t()[None], "image_size": size[None]}})
            return
        lightglue_ref.lightglue_forward(f[0]["keypoints"], f[0]["descriptors"].t().contiguous(), size,
                                        f[1]["keypoints"], f[1]["descriptors"].t().contiguous(), size, lg_sd, conf)

    one_pair(1000)  # warm-up
    times = []
    for i in range(n_pairs):
        t0 = time.perf_counter()
        one_pair(i)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2] if len(times) % 2 else 0.5 * (sorted(times)[len(times) // 2 - 1] + sorted(times)[len(times) // 2])
    kind = "reference" if ref_mods is not None else "port"
    what = "the reference's SuperPoint / LightGlue modules (imported from /root/reference)" if ref_mods is not None else "oracle/*.py"
    return {"value": 1.0 / med, "unit": "image-pairs/s", "cores": cores, "kind": kind, "statistic": f"1 / median of {n_pairs} per-pair times (SURVEY 8(d): warm-up 1, median of >= 5)",
            "mean_value": n_pairs / sum(times), "per_pair_s": [round(t, 3) for t in times],
            "sample": f"{n_pairs} pairs (= {2 * n_pairs} SuperPoint 1024x1024 forwards + {n_pairs} LightGlue 2048x2048 "
                      f"9-layer forwards) after 1 warm-up pair, {what} on torch CPU, {cores} threads"}


def measure_config4(a, rank, world, dev, dist, lib, steps, warmup):
    """BASELINE configs[3] (SURVEY 8(d) "config 4"): `--images` synthetic 1024^2 images -> the first `--job-pairs` exhaustive
    pairs (pairs_generator.py:37-38 order) through pipeline.PairMatchingPipeline: images sharded i mod world, ONE all-gather of
    the feature tables, pairs dealt by cost n0 x n1 (pipeline.balanced_shards), ONE all-gather of the match tables.  STRONG scaling: the job is fixed.  A
    "step" is one pass over the whole job; per-phase wall times are the max over ranks.  The images are crops of one canvas
    (workloads.shifted_crops) and LightGlue runs the matching-capable synthetic weights at the reference's default threshold 0.1,
    so the match tables that cross xGMI are filled with hundreds of true correspondences per pair.
    Returns the record (rank 0) or None."""
    sp = importlib.import_module(PKG + ".superpoint_hip")
    lg = importlib.import_module(PKG + ".lightglue_hip")
    pl = importlib.import_module(PKG + ".pipeline")
    weights = importlib.import_module(PKG + ".weights")
    workloads = importlib.import_module(PKG + ".workloads")
    capi = importlib.import_module(PKG + ".capi")
    B, K, W = a.pairs, steps, warmup
    # the job's geometry: BASELINE's (1024 x 1024 images, 2048 keypoints) unless the caller scales it down — tests/test_bench_multirank_gloo.py runs
    # this very function at world 8 over gloo on the CPU emulator build with 48 x 64 images and 32 keypoints
    (IH, IW), NKP = getattr(a, "c4_hw", (1024, 1024)), int(getattr(a, "c4_kpts", 2048))
    if os.environ.get("DIM_BENCH_TEST_FAIL_RANK") == str(rank):     # test hook: a failure on ONE rank only (the watchdog / error path of main())
        raise RuntimeError(f"injected failure on rank {rank}")
    on_gpu = torch.device(dev).type == "cuda"
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": NKP, "remove_borders": 4}
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
    ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=B, max_hw=(IH, IW), capacity=NKP, device=dev, lib=None if on_gpu else lib)
    imgs = workloads.shifted_crops(a.images, IH, IW, max_shift=min(256, IH // 4 // 8 * 8), seed=7)[0].to(dev)
    center = workloads.descriptor_mean(ext, imgs)     # same on every rank (same images, deterministic kernels)
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_matching_state_dict(0, 256, center=center), conf, max_pairs=B, max_kpts=NKP, device=dev,
                          lib=None if on_gpu else lib)
    pipe = pl.PairMatchingPipeline(ext, mat, rank, world)
    pairs = pl.exhaustive_pairs(a.images, a.job_pairs)
    P = int(pairs.shape[0])

    def barrier():
        _sync(dev)
        if dist is not None:
            dist.barrier()
        _sync(dev)

    for _ in range(W):   # reduced pass: both networks, both collectives, every buffer size class touched once
        nw = min(a.images, 2 * B * world)
        pipe.match_all(pipe.extract_all(imgs[:nw]), pl.exhaustive_pairs(nw, B * world))
    barrier()
    capi.check(lib, lib.dim_profile_start(ctypes.c_ulonglong(1 << 13)))   # DIM_PROF_LG_SELF_ATTN: the self-attention launches
    phases = {"extract_s": 0.0, "feature_gather_s": 0.0, "match_s": 0.0, "match_gather_s": 0.0}
    t0 = time.perf_counter()
    total_matches = 0
    for _ in range(K):
        table = pipe.extract_all(imgs)
        cnt, mt, ms = pipe.match_all(table, pairs)
        for k in phases:
            phases[k] += pipe.timings[k]
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    capi.check(lib, lib.dim_profile_stop(ctypes.byref(tot_ms), ctypes.byref(launches)))
    total_matches = int(cnt.sum().item())
    pairs_with_100 = int((cnt >= 100).sum().item())
    tt = torch.tensor([dt] + [phases[k] for k in phases], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0].item())
    sat_total, sat_sites = capi.saturation(lib, _stream_ptr(dev), reset=True)
    gathered = {"feature_gather_bytes": pipe.timings.get("feature_gather_bytes"), "match_gather_bytes": pipe.timings.get("match_gather_bytes")}
    del ext, mat, pipe, imgs, table, mt, ms
    if on_gpu:
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    per_rank_pairs = (P + world - 1) // world
    attn_ms = tot_ms.value / max(1, launches.value)
    # one self-attention launch = 2B items x 4 heads: QK^T + AV = 77.3 GFLOP per pair over 9 layers; the last
    # batch of a shard may be smaller, so the algorithmic work per launch is averaged over the launches actually made
    gflop_per_launch = 77.3 * per_rank_pairs * K / max(1, launches.value)
    return {
        "metric": "image-pairs/s (SuperPoint+LightGlue, 1024^2, 2048 kpts)", "value": K * P / dt, "unit": "image-pairs/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[3] (config 4): {a.images} synthetic {IH}x{IW} images (crops of one canvas: true correspondences) -> first {P} "
                               "exhaustive pairs, extraction amortised over the images + 1 LightGlue match per pair (fixed-work, 9 layers, matching-capable "
                               "synthetic weights, threshold 0.1), through PairMatchingPipeline phases 1-4",
                   "images": a.images, "job_pairs": P, "pair_batch": B, "gflop_per_pair": LG_GFLOP_PER_PAIR + SP_GFLOP_PER_IMAGE * a.images / P,
                   "sharding": f"images i mod {world}, ONE all-gather of feature tables ({pipe_bytes(a.images, world, NKP, 256, NKP)[0] / 1e6:.0f} MB), pairs "
                               f"dealt by cost n0 x n1 (balanced_shards; equal costs = round-robin), ONE all-gather of match tables ({pipe_bytes(P, world, NKP, 256, NKP)[1] / 1e6:.0f} MB)"},
        "gathered_bytes_measured": gathered,
        "phases_s_max_over_ranks": {k: float(v) / K for k, v in zip(phases, tt[1:].tolist())},
        "matches_total": total_matches, "matches_per_pair_mean": total_matches / max(1, P), "pairs_with_at_least_100_matches": pairs_with_100,
        "fp16x3_range_guard": {"violations": sat_total, "sites": sat_sites},
        "roofline": {"kernel": "attn_x6_kernel<2> self-attention launches (flash attention, fp16x3 on the fp16 MFMA)", "bound": "mfma",
                     "achieved": gflop_per_launch / attn_ms, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": gflop_per_launch / attn_ms / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "avg_launch_ms": attn_ms,
                     "launches": launches.value, "algorithmic_gflop_per_launch": gflop_per_launch},
        "cpu_baseline": None,
    }


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def _stream_ptr(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if torch.device(dev).type == "cuda" else None


def pipe_bytes(n_items, world, cap=2048, D=256, NK=2048):
    """(feature gather bytes for n_items images, match gather bytes for n_items pairs): the flat buffers of pipeline.py x world"""
    per = (n_items + world - 1) // world
    return (per * cap * (2 + 1 + D) + per) * 4 * world, (per * 2 + per * NK * 3) * 4 * world


def run_config4(a, rank, world, dev, dist, lib):
    line = measure_config4(a, rank, world, dev, dist, lib, a.steps, a.warmup)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_config5(a, rank, world, dev, dist, lib):
    """BASELINE configs[4] (SURVEY 8(d) "config 5"): ALIKED (aliked-n16rot geometry, 4000 keypoints per 1500 x 1000 tile, 4 x 4 tiles) +
    LightGlue (128-d) on `--images` synthetic 6000 x 4000 RGB images -> all exhaustive image pairs, through pipeline.TiledPairPipeline:
    images i mod world -> batched _extract_by_tile -> ONE feature all-gather -> pairs j mod world -> tile preselection (down-sampled
    SuperPoint + LightGlue on the device) + batched tile-pair matching -> ONE match all-gather.  STRONG scaling.  The images are crops of
    one canvas (offsets multiples of 32 px = ALIKED's total stride) and both LightGlue instances run matching-capable synthetic
    weights, so tile pairs ARE selected and the match tables are filled.  Roofline object: ALIKED's two full-resolution 3 x 3
    convolutions (HBM-bound: algorithmic bytes = fp32 input + output maps of one launch), timed with HIP events on the launch stream."""
    import numpy as np
    plugins = importlib.import_module(PKG + ".plugins")
    pl = importlib.import_module(PKG + ".pipeline")
    tm = importlib.import_module(PKG + ".tile_matching")
    weights = importlib.import_module(PKG + ".weights")
    capi = importlib.import_module(PKG + ".capi")
    K, W = a.steps, a.warmup
    # tile_preselection_size 750 (the reference's default is 1024): 6000 x 4000 images are then down-sampled by EXACTLY 8 (INTER_AREA = 8 x 8 box
    # means), so crops of one canvas at multiples of 64 px have bit-identical down-sampled overlaps shifted by multiples of 8 px — the shifts
    # SuperPoint is equivariant under — and the preselector's seeded SuperPoint + matching-capable LightGlue find the true correspondences:
    # PRESELECTION votes and selects the tile pairs itself, nothing falls back to GRID
    PRE = 750
    general = {"tile_size": (1500, 1000), "tile_overlap": 0, "tile_preselection_size": PRE, "min_matches_per_tile": 5, "quality": "HIGH",
               "allow_synthetic_weights": True}
    ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 4000,
                                                                     "detection_threshold": 0.2, "nms_radius": 3, "allow_synthetic_weights": True}})
    mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", "depth_confidence": 0.95, "width_confidence": 0.99,
                                                                   "filter_threshold": 0.1, "allow_synthetic_weights": True}}, local_features="aliked")
    mt._sd = weights.synthetic_lightglue_matching_state_dict(0, 128)          # ALIKED's synthetic descriptors are discriminative (mean cosine 0.12): no centring
    rng = np.random.default_rng(5)
    canvas = rng.integers(0, 256, (4000 + 512, 6000 + 512, 3), dtype=np.uint8)
    # the FIRST band (the one tile_selection reads, MB:1021-1024) is noise at 8 x 8-block scale: its down-sampled image keeps the full contrast, which the
    # seeded SuperPoint needs for distinguishable descriptors (pixel-scale noise box-averages to a flat grey: 3500 common keypoints, 4 matches)
    canvas[..., 0] = np.kron(rng.integers(0, 256, ((4000 + 512) // 8, (6000 + 512) // 8), dtype=np.uint8), np.ones((8, 8), np.uint8))
    offs = [(0, 0)] + [(int(rng.integers(0, 9)) * 64, int(rng.integers(0, 9)) * 64) for _ in range(a.images - 1)]      # multiples of 64 (and so of ALIKED's stride 32)
    images = [np.ascontiguousarray(canvas[dy:dy + 4000, dx:dx + 6000]).astype(np.float32) for dy, dx in offs]
    del canvas
    # the preselector (SuperPoint + LightGlue at 1024 px) with matching-capable weights centred on its own descriptors
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    pre = tm.TilePreselector(sp_sd, weights.synthetic_lightglue_state_dict(0, 256), tile_preselection_size=PRE, device=dev, lib=lib)
    f0 = pre.features("warm", np.ascontiguousarray(images[0][..., 0]), "HIGH")
    center = f0[1][0, : int(f0[2][0])].mean(0).cpu()
    mt._tile_preselector = tm.TilePreselector(sp_sd, weights.synthetic_lightglue_matching_state_dict(0, 256, center=center), tile_preselection_size=PRE,
                                              device=dev, lib=lib)
    del pre
    pipe = pl.TiledPairPipeline(ex, mt, rank, world, selection=a.tile_selection, empty_selection_fallback="GRID" if a.tile_selection.startswith("PRESELECTION") else None,
                                tile_pair_batch=a.tile_pair_batch)
    pairs = pl.exhaustive_pairs(a.images, a.job_pairs)
    P = int(pairs.shape[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):    # reduced pass: one image per rank, one pair per rank
        nw = min(a.images, max(2, world))
        fw = pipe.extract_all(images[:nw])
        pipe.match_all(images[:nw], fw, pl.exhaustive_pairs(nw, world))
    barrier()
    pipe.n_fallback = 0
    capi.check(lib, lib.dim_profile_start(ctypes.c_ulonglong(1 << 17)))     # DIM_PROF_AL_CONV_FULL
    phases = {"extract_s": 0.0, "feature_gather_s": 0.0, "match_s": 0.0, "tile_selection_s": 0.0, "tile_matching_s": 0.0, "match_gather_s": 0.0}
    t0 = time.perf_counter()
    for _ in range(K):
        mt._tile_preselector._cache.clear()
        feats = pipe.extract_all(images, as_numpy=False)
        matches = pipe.match_all(images, feats, pairs)
        for k in phases:
            phases[k] += pipe.timings[k]
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    capi.check(lib, lib.dim_profile_stop(ctypes.byref(tot_ms), ctypes.byref(launches)))
    tt = torch.tensor([dt] + [phases[k] for k in phases], dtype=torch.float64, device=dev)
    fb = torch.tensor([pipe.n_fallback, pipe.timings.get("tile_pairs_this_rank", 0)], dtype=torch.int64, device=dev)
    cost = torch.tensor([pipe.timings.get("cost_this_rank", 0.0)], dtype=torch.float64, device=dev)
    cost_max = cost.clone()
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
        dist.all_reduce(cost_max, op=dist.ReduceOp.MAX)
    dt = float(tt[0].item())
    sat_total, sat_sites = capi.saturation(lib, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream), reset=True)
    if rank == 0:
        conv_ms = tot_ms.value / max(1, launches.value)
        # one launch = a batch of 16 tiles padded to 1024 x 1504 (multiples of 32); block1.conv1 reads 3 and writes 16 fp32 channels,
        # block1.conv2 reads 16 and writes 16: the average launch moves (19 + 32) / 2 channels x 4 B per pixel
        px = 16 * 1024 * 1504
        bytes_per_launch = px * (19 + 32) / 2 * 4
        nm = [int(m.shape[0]) for m in matches]
        line = {
            "metric": "image-pairs/s (ALIKED+LightGlue, tiled 6000x4000, 4x4 tiles, 4000 kpts per tile)", "value": K * P / dt, "unit": "image-pairs/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4] (config 5): {a.images} synthetic 6000x4000 RGB images (crops of one canvas) -> {P} exhaustive image pairs; ALIKED "
                                   "aliked-n16rot geometry, 16 tiles of 1500x1000 per image, 4000 keypoints per tile; tile selection " + a.tile_selection + " on the device; LightGlue "
                                   "(128-d, adaptive depth / width 0.95 / 0.99, threshold 0.1) on the selected tile pairs; seeded synthetic weights",
                       "images": a.images, "job_pairs": P,
                       "sharding": f"images i mod {world}, ONE all-gather of the merged tile tables ({pipe.timings['feature_gather_bytes'] / 1e6:.0f} MB), tile selection of image pairs j mod {world}, "
                                   f"ONE small all-gather of the selection masks ({pipe.timings.get('selection_gather_bytes', 0)} B), image pairs dealt by cost, "
                                   f"ONE all-gather of the match rows ({pipe.timings['match_gather_bytes'] / 1e6:.0f} MB)"},
            "phases_s_max_over_ranks": {k: float(v) / K for k, v in zip(phases, tt[1:].tolist())},
            "keypoints_per_image_mean": float(np.mean([int(f["keypoints"].shape[0]) for f in feats])),
            "matches_per_pair_mean": float(np.mean(nm)), "matches_per_pair_min": int(min(nm)), "pairs_with_matches": int(sum(1 for x in nm if x > 0)),
            "tile_pairs_total": int(pipe.timings.get("tile_pairs_total", 0)), "tile_pairs_per_s": float(pipe.timings.get("tile_pairs_total", 0)) * K / dt,
            "tile_pairs_per_image_pair_mean": float(pipe.timings.get("tile_pairs_total", 0)) / max(1, P),
            "image_pairs_fell_back_to_grid": int(fb[0].item()) // max(1, K),
            "deal_balance": {"cost_max_over_ranks": float(cost_max.item()), "cost_mean_over_ranks": float(pipe.timings.get("cost_total", 0.0)) / world,
                             "note": "image pairs dealt by sum(n0 x n1) over the SELECTED tile pairs (pipeline.balanced_shards)"},
            "tile_selection_note": f"{a.tile_selection} on the device for every image pair (down-sampled SuperPoint + matching-capable LightGlue, tile_preselection_size {PRE} = "
                                   "an exact 8 x 8 box down-sampling of 6000 x 4000, crops at multiples of 64 px): the votes select the tile pairs — a tile of image 0 "
                                   "overlaps up to 4 tiles of image 1 —, nothing falls back",
            "fp16x3_range_guard": {"violations": sat_total, "sites": sat_sites},
            "roofline": {"kernel": "al_convx3_kernel<16, 9, 16> full-resolution launches (ALIKED block1.conv1 3 -> 16 and block1.conv2 16 -> 16, fp16x3, BatchNorm "
                                   "statistics in the epilogue)", "bound": "hbm", "achieved": bytes_per_launch / conv_ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                         "frac": bytes_per_launch / conv_ms / 1e6 / 8000.0, "traffic": None, "avg_launch_ms": conv_ms, "launches": launches.value,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "cpu_baseline": None,
        }
        if world == 1 and not a.no_live_traffic:
            # after everything that is timed: two counter passes over a child that runs the same extraction launches (16 tiles of 1000 x 1500 per call)
            try:
                tr = measure_kernel_traffic([sys.executable, str(ROOT / "scripts" / "gpu_aliked_one.py"), "16", "1000", "1500", "2"], "al_convx3_kernel<16, 9, 16")
                r = line["roofline"]
                r["traffic"] = tr["hbm_bytes_per_launch"]
                r["traffic_measurement"] = tr
                r["traffic_note"] = ("MEASURED by this run: HBM bytes per full-resolution launch of this kernel (mean of block1.conv1 and block1.conv2) from two separate "
                                     "rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE) over scripts/gpu_aliked_one.py 16 1000 1500, FETCH_SIZE doubled")
                r["traffic_over_algorithmic"] = tr["hbm_bytes_per_launch"] / bytes_per_launch
                r["frac_of_hbm_on_measured_traffic"] = tr["hbm_bytes_per_launch"] / conv_ms / 1e6 / 8000.0
            except Exception as e:
                line["roofline"]["traffic_measurement"] = {"error": repr(e)[:300]}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def config1_cpu_leg(repeats: int = 1):
    """configs[0] as BASELINE.json words it: the reference's CPU path (general.force_cpu) on the five sacre-coeur photographs -> 10 brute-force
    pairs, config/superpoint+lightglue.yaml.  The reference's SuperPoint / LightGlue modules themselves when /root/reference exists (kind
    "reference": the build container), else the oracle (kind "port": the GPU box).  Decode (PIL) + Q5 grey + _frame2tensor + SuperPoint forward per
    image; float16 round trip (features.h5) + featuresDict2Lightglue + LightGlue forward per pair; no file IO, no RANSAC."""
    import numpy as np
    from oracle import lightglue_ref, superpoint_ref
    from tests import golden_cases as gc

    weights = importlib.import_module(PKG + ".weights")
    cores = min(16, os.cpu_count() or 16)
    torch.set_num_threads(cores)
    sp_sd = weights.synthetic_superpoint_state_dict(1234)
    lg_sd = weights.synthetic_lightglue_state_dict(0, 256, gain=2.0)
    cfg, conf = dict(gc.CONFIG1_SP), dict(gc.CONFIG1_LG)
    ref_mods = _reference_modules()
    if ref_mods is not None:
        spn, lgn = ref_mods
        orig = torch.hub.load_state_dict_from_url
        torch.hub.load_state_dict_from_url = lambda *a_, **k_: sp_sd
        try:
            ref_sp = spn.SuperPoint({k: v for k, v in cfg.items() if k != "fix_sampling"}).eval()
        finally:
            torch.hub.load_state_dict_from_url = orig
        ref_lg = lgn.LightGlue(features=None, input_dim=256, **conf).eval()
        ref_lg.load_state_dict(lg_sd, strict=False)
    grays = [gc.real_gray(n) for n in gc.SACRE_COEUR]

    @torch.no_grad()
    def job():
        t0 = time.perf_counter()
        feats = []
        for g in grays:
            img = torch.tensor(g[None][None] / 255.0, dtype=torch.float)
            if ref_mods is not None:
                o = ref_sp({"image": img})
                f = {"keypoints": o["keypoints"][0].numpy(), "scores": o["scores"][0].numpy(), "descriptors": o["descriptors"][0].numpy()}
            else:
                o = superpoint_ref.superpoint_forward(img, sp_sd, cfg)
                f = {k: o[k].numpy() for k in ("keypoints", "scores", "descriptors")}
            f = gc.fp16_round_trip(f)
            f["size"] = torch.tensor(g.shape[:2], dtype=torch.float32)
            feats.append(f)
        t1 = time.perf_counter()
        n_matches = 0
        for a, b in gc.config1_pairs():
            fa, fb = feats[a], feats[b]
            ka, kb = torch.tensor(fa["keypoints"]), torch.tensor(fb["keypoints"])
            da, db = torch.tensor(fa["descriptors"]).t().contiguous(), torch.tensor(fb["descriptors"]).t().contiguous()
            if ref_mods is not None:
                r = ref_lg({"image0": {"keypoints": ka[None], "descriptors": da[None], "image_size": fa["size"][None]},
                            "image1": {"keypoints": kb[None], "descriptors": db[None], "image_size": fb["size"][None]}})
                n_matches += int(r["matches"][0].shape[0])
            else:
                r = lightglue_ref.lightglue_forward(ka, da, fa["size"], kb, db, fb["size"], lg_sd, conf)
                n_matches += int(r["matches"].shape[0])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, n_matches

    job()   # warm-up
    runs = [job() for _ in range(max(1, repeats))]
    ext_s, match_s = min(r[0] for r in runs), min(r[1] for r in runs)
    kind = "reference" if ref_mods is not None else "port"
    return {"value": 10 / (ext_s + match_s), "unit": "image-pairs/s", "cores": cores, "kind": kind, "extract_s": ext_s, "match_s": match_s,
            "ms_per_image": ext_s / 5 * 1e3, "ms_per_pair": match_s / 10 * 1e3, "matches_total": runs[0][2],
            "sample": f"the whole config-1 job (5 SuperPoint forwards at 640x480 .. 784x784 + 10 LightGlue forwards at 2000 x 2000 keypoints, adaptive depth / width), best of "
                      f"{max(1, repeats)} after 1 warm-up job, " + ("the reference's SuperPoint / LightGlue modules (imported from /root/reference)" if ref_mods is not None else "oracle/*.py")
                      + f" on torch CPU, {cores} threads"}


def run_config1(a, dev, lib):
    """BASELINE configs[0] on its real inputs: the five photographs of assets/example_sacre_coeur (tests/assets/config1: byte copies) -> 10 brute-force
    pairs, config/superpoint+lightglue.yaml (nms 4 / thr 0.005 / 2000 keypoints; LightGlue 0.95 / 0.99 / 0.1), seeded synthetic weights (the trained
    files are URL downloads).  `value` = pairs/s of the whole job through the PER-CALL PLUGIN HOOKS (what the reference's own loops get when the plugins
    are dropped in: one image / one pair per call, host arrays in and out, one guard read-back per call); sub-records: the same job through
    BatchedImageMatcher (files -> features.h5 -> raw_matches.h5) and the CPU leg (`cpu_baseline`)."""
    import tempfile
    import numpy as np
    from tests import golden_cases as gc

    if a.cpu_only:
        print(json.dumps({"metric": "image-pairs/s (SuperPoint+LightGlue, config 1: 5 sacre-coeur photographs -> 10 pairs)", "value": None, "unit": "image-pairs/s",
                          "n_gpus": 0, "config": {"workload": "configs[0] (config 1), CPU leg only"}, "cpu_baseline": config1_cpu_leg(repeats=3)}))
        return
    plugins = importlib.import_module(PKG + ".plugins")
    bm = importlib.import_module(PKG + ".batched_matcher")
    capi = importlib.import_module(PKG + ".capi")
    K, W = a.steps, a.warmup
    ex = plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", **gc.CONFIG1_SP, "allow_synthetic_weights": True}})
    mt = plugins.LightGlueMatcher({"general": {"geom_verification": "NONE"}, "matcher": {"name": "lightglue", **gc.CONFIG1_LG, "allow_synthetic_weights": True}},
                                  local_features="superpoint")
    mt._sd = importlib.import_module(PKG + ".weights").synthetic_lightglue_state_dict(0, 256, gain=2.0)
    grays = [gc.real_gray(n) for n in gc.SACRE_COEUR]
    sizes = [np.array(g.shape[:2], dtype=np.int32) for g in grays]

    def hook_job():
        t0 = time.perf_counter()
        feats, rt = [], 0.0
        for g, sz in zip(grays, sizes):
            raw = ex._extract(g)
            ta = time.perf_counter()
            f = gc.fp16_round_trip(raw)       # ExtractorBase.extract -> save_features_h5 -> get_features (Q6): the reference's own float16 casts, on the host
            rt += time.perf_counter() - ta
            f["image_size"] = sz
            feats.append(f)
        t1 = time.perf_counter()
        n = 0
        for i, j in gc.config1_pairs():
            n += int(mt._match_pairs(feats[i], feats[j]).shape[0])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, n, rt

    for _ in range(max(1, W)):
        hook_job()
    runs = [hook_job() for _ in range(K)]
    ext_s, match_s = float(np.median([r[0] for r in runs])), float(np.median([r[1] for r in runs]))
    paths = [gc.REAL_DIR / n for n in gc.SACRE_COEUR]
    pairs = [(paths[i].name, paths[j].name) for i, j in gc.config1_pairs()]
    batched = []
    for it in range(W + K):
        with tempfile.TemporaryDirectory() as d:
            shim = bm.BatchedImageMatcher(ex, mt, d, image_batch=4, pair_batch=10, verify=False)
            t0 = time.perf_counter()
            fpath = shim.extract_features(paths)
            t1 = time.perf_counter()
            shim.match_pairs(fpath, pairs)
            t2 = time.perf_counter()
        if it >= W:
            batched.append((t1 - t0, t2 - t1))
    b_ext, b_match = float(np.median([r[0] for r in batched])), float(np.median([r[1] for r in batched]))
    sat_total, sat_sites = capi.saturation(lib, None, reset=True)
    line = {
        "metric": "image-pairs/s (SuperPoint+LightGlue, config 1: 5 sacre-coeur photographs -> 10 pairs)", "value": 10 / (ext_s + match_s), "unit": "image-pairs/s",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": (ext_s + match_s) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "real photographs (assets/example_sacre_coeur), seeded synthetic weights",
        "config": {"workload": "configs[0] (config 1): 5 photographs (640x480, 618x640 x2, 640x618, 784x784) -> 10 brute-force pairs; config/superpoint+lightglue.yaml "
                               "(SuperPoint nms 4 / thr 0.005 / 2000 keypoints; LightGlue depth 0.95 / width 0.99 / threshold 0.1); one step = the whole job through the per-call "
                               "plugin hooks (_extract x 5, float16 round trip, _match_pairs x 10), host arrays in and out", "images": 5, "pairs": 10},
        "hook_path": {"extract_s": ext_s, "match_s": match_s, "ms_per_image": ext_s / 5 * 1e3, "ms_per_pair": match_s / 10 * 1e3, "matches_total": runs[0][2],
                      "float16_round_trip_s": float(np.median([r[3] for r in runs])),
                      "note": "extract_s includes float16_round_trip_s: numpy's float32 -> float16 -> float32 casts that stand in for save_features_h5 / get_features "
                              "(the reference's own host work between the two hooks); _extract alone = (extract_s - float16_round_trip_s) / 5"},
        "batched_image_matcher": {"value": 10 / (b_ext + b_match), "unit": "image-pairs/s", "extract_features_s": b_ext, "match_pairs_s": b_match,
                                  "note": "JPEG files -> PIL decode + Q5 grey -> batched dim_sp_extract (images bucketed by shape) -> features store (float16, deflate) -> "
                                          "re-read -> ONE dim_lg_match of the 10 pairs -> raw_matches store; file IO included"},
        "fp16x3_range_guard": {"violations": sat_total, "sites": sat_sites},
        "roofline": None,
        "cpu_baseline": None if a.no_cpu_baseline else config1_cpu_leg(repeats=1),
    }
    print(json.dumps(line))


def measure_hook_path(dev, lib, n_img: int = 12, n_pair: int = 24):
    """What a user who drops the plugin classes into the reference's own loops gets (image_matching.py:429-430, 467-487: one image / one pair per call):
    SuperPointExtractor._extract on a 1024 x 1024 float32 image (numpy in, numpy out: H2D, ~25 launches, guard read-back, D2H) and
    LightGlueMatcher._match_pairs on two 2048-keypoint numpy feature dicts, fixed work (9 layers) — batch 1, wall time per call."""
    import numpy as np
    plugins = importlib.import_module(PKG + ".plugins")
    ex = plugins.SuperPointExtractor({"general": {}, "extractor": {"name": "superpoint", "nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048,
                                                                   "remove_borders": 4, "allow_synthetic_weights": True}})
    mt = plugins.LightGlueMatcher({"general": {}, "matcher": {"name": "lightglue", "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1,
                                                               "allow_synthetic_weights": True}}, local_features="superpoint")
    g = torch.Generator().manual_seed(77)
    imgs = [(torch.rand(1024, 1024, generator=g) * 255).numpy().astype(np.float32) for _ in range(2)]
    feats = []
    for im in imgs:                      # warm-up + the features of the pair
        f = ex._extract(im)
        f["image_size"] = np.array([1024, 1024], dtype=np.int32)
        feats.append(f)
    for _ in range(3):
        mt._match_pairs(feats[0], feats[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_img):
        ex._extract(imgs[i % 2])
    t1 = time.perf_counter()
    for i in range(n_pair):
        mt._match_pairs(feats[0], feats[1])
    t2 = time.perf_counter()
    # the same two calls without the host hand-over: device tensors in, device tensors out, batch 1 (kernel time of the call)
    net, lgn = ex._net, mt._net
    img_d = torch.from_numpy(imgs[0] / 255.0).to(dev)[None].contiguous()
    out_sp = net.extract_batch(img_d)
    kp, sc, de, n = out_sp
    kt = torch.stack([kp[0], kp[0]]).contiguous(); dt_ = torch.stack([de[0], de[0]]).contiguous()
    nt = torch.stack([n[0], n[0]]).contiguous(); st = torch.full((2, 2), 1024.0, device=dev)
    out_lg = lgn.match_batch(kt, dt_, nt, st, n_pairs=1)
    for _ in range(3):
        lgn.match_batch(kt, dt_, nt, st, n_pairs=1, out=out_lg)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(n_img):
        net.extract_batch(img_d, out=out_sp)
    e[1].record()
    for _ in range(n_pair):
        lgn.match_batch(kt, dt_, nt, st, n_pairs=1, out=out_lg)
    e[2].record()
    torch.cuda.synchronize()
    rec = {"ms_per_image": (t1 - t0) / n_img * 1e3, "ms_per_pair": (t2 - t1) / n_pair * 1e3,
           "pairs_per_s_extract2_match1": 1e3 / (2 * (t1 - t0) / n_img * 1e3 + (t2 - t1) / n_pair * 1e3),
           "device_only_ms_per_image": e[0].elapsed_time(e[1]) / n_img, "device_only_ms_per_pair": e[1].elapsed_time(e[2]) / n_pair,
           "note": "batch-1 calls through plugins.SuperPointExtractor._extract / LightGlueMatcher._match_pairs (numpy in / out, one guard read-back per call) at the "
                   "headline sizes (1024 x 1024, 2048 x 2048 keypoints, 9 layers); device_only_* = the same batch-1 library calls on resident tensors (HIP events). "
                   "The calls are kernel-bound, not launch-bound: the rocprofv3 kernel times of a call add up to its wall time (profiles/r06_hook_path_b1_kernel_stats.csv; round 6: 32 x 128 GEMM blocks, chunk-ahead requests, K | V images from the small projection — profiles/r06_b1_steps.json; adaptive pairs: profiles/r06_b1_adaptive.json)"}
    del ex, mt
    torch.cuda.empty_cache()
    return rec


def measure_adaptive(dev, lib, P: int = 50, reps: int = 5):
    """The reference's DEFAULT LightGlue (adaptive depth 0.95 / width 0.99, LGN:494-516, 586-604) where it adapts: a
    50-pair x 2048-keypoint batch whose pairs stop after 3 .. 9 layers and lose a quarter of their keypoints to pruning after the first layer
    (workloads.adaptive_lightglue_workload: one set of matching-capable weights; the stop layers are designed into the inputs), timed next
    to the fixed-work call on the SAME inputs.  `ideal` = the matrix work the adaptive call really needs / the fixed-work call's (per layer and
    pair: the MACs of SURVEY 8(d)'s formula at the keypoint counts that are still alive), so achieved_of_ideal = (t_fixed x ideal) / t_adaptive."""
    lg = importlib.import_module(PKG + ".lightglue_hip")
    wl = importlib.import_module(PKG + ".workloads")
    N = 2048
    sd, kp, de, nt, st, expect = wl.adaptive_lightglue_workload(P, N)
    kp, de, nt, st = kp.to(dev), de.to(dev), nt.to(dev), st.to(dev)
    rec = {}
    outs = {}
    for name, conf in (("fixed_work", {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}),
                       ("reference_default", {"depth_confidence": 0.95, "width_confidence": 0.99, "filter_threshold": 0.1, "pruning_min_kpts": 1536})):
        net = lg.LightGlueHIP(sd, conf, max_pairs=P, max_kpts=N, device=dev)
        out = net.match_batch(kp, de, nt, st)
        for _ in range(2):
            net.match_batch(kp, de, nt, st, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            net.match_batch(kp, de, nt, st, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rec[name] = {"ms_per_batch": ms, "pairs_per_s": P / ms * 1e3, "matches_mean": float(out["n_matches"].float().mean())}
        outs[name] = {k: out[k].cpu() for k in ("stop", "prune01", "n_matches")}
        del net, out
        torch.cuda.empty_cache()
    stop = outs["reference_default"]["stop"].long()
    prune = outs["reference_default"]["prune01"].long()          # [P, 2, N]: layers a keypoint took part in

    def macs(n0, n1):   # one layer of one pair with n0 / n1 live keypoints (self blocks, cross block)
        lin = lambda n: n * 256 * 768 + n * 256 * 256 + n * 512 * 512 + n * 512 * 256            # noqa: E731  q|k|v, out, ffn.0, ffn.3
        lin_c = lambda n: 3 * n * 256 * 256 + n * 512 * 512 + n * 512 * 256                        # noqa: E731  to_qk, to_v, out, ffn
        return lin(n0) + lin(n1) + 2 * 256 * (n0 * n0 + n1 * n1) + lin_c(n0) + lin_c(n1) + 3 * 256 * n0 * n1

    # alive at layer l: the counter (LGN:509,516: +1 per pruning step the keypoint survived) exceeds l — or equals the side's maximum: below
    # pruning_min_kpts the reference stops pruning AND counting, so the survivors' counters stand still
    need = 0.0
    for p_ in range(P):
        mx = prune[p_].max(-1).values
        for l in range(int(stop[p_])):
            need += macs(int(((prune[p_, 0] > l) | (prune[p_, 0] == mx[0])).sum()), int(((prune[p_, 1] > l) | (prune[p_, 1] == mx[1])).sum()))
    full = P * 9 * macs(N, N)
    hist = {str(k): int((stop == k).sum()) for k in sorted(set(stop.tolist()))}
    t_fix, t_ad = rec["fixed_work"]["ms_per_batch"], rec["reference_default"]["ms_per_batch"]
    rec.update({"stop_layer_histogram": hist, "stop_layers_as_designed": bool(torch.equal(stop, expect.long())),
                "keypoints_alive_after_layer_1_mean": float((prune > 1).sum(-1).float().mean()),
                "kernel_level": "profiles/r06_adaptive_kernel_stats_{adaptive,fixed}.csv (scripts/gpu_adaptive_trace.sh): attention 0.47 of its fixed-work time (ideal 0.42), "
                                "fused feed-forward 0.62 and q|k|v 0.64 (ideal 0.52: a stopped pair's and a pruned row's workgroups still launch and leave at once); "
                                "token confidence + prune scan / gather / copy-back / commit + decide = 0.83 ms per batch on top",
                "ideal_depth_only": float(stop.float().mean() / 9.0), "ideal": need / full,
                "adaptive_over_fixed_time": t_ad / t_fix, "achieved_of_ideal": (t_fix * need / full) / t_ad,
                "workload": f"{P} pairs x {N} x {N} keypoints, workloads.adaptive_lightglue_workload (stop layers 3..9 in equal shares, 25 % of the keypoints prunable "
                            "after layer 1), matching-capable weights, filter_threshold 0.1; both calls on the same resident inputs, HIP events over "
                            f"{reps} calls"})
    return rec


def measure_kernel_traffic(child, kernel_substr: str, timeout_s: float = 240.0):
    """HBM bytes per launch of one kernel from the L2's memory-side counters, collected as MI355X_MICROARCH.md section HBM prescribes — two
    SEPARATE rocprofv3 passes (`--kernel-trace --pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; no trace domain next to the counters), each over the
    child command `child`, FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes), WRITE_SIZE as
    reported (it checks out exactly against known byte counts), counter unit KiB per dispatch; averaged over the launches with the LARGEST grid
    whose kernel name contains `kernel_substr`.  Returns {"hbm_bytes_per_launch", "fetch_bytes_corrected_x2", "write_bytes", "launches_counted",
    "seconds"}."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise RuntimeError("rocprofv3 not on PATH")
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="dim_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", *child]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=timeout_s / 2)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)          # exactly the process group this call started
                raise RuntimeError(f"the {counter} pass did not finish in {timeout_s / 2:.0f} s")
            files = [os.path.join(r, f) for r, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if p.returncode != 0 or not files:
                raise RuntimeError(f"the {counter} pass failed (rc {p.returncode}, {len(files)} counter files)")
            rows = [r for r in csv.DictReader(open(files[0])) if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not rows:
                raise RuntimeError(f"no dispatch of the kernel in the {counter} pass")
            grid = max(int(r["Grid_Size"]) for r in rows)       # the full-size launches (a warm-up step has the same shape)
            vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == grid]
            per[counter] = (sum(vals) / len(vals) * 1024.0, len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = 2.0 * per["FETCH_SIZE"][0], per["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
            "launches_counted": {"FETCH_SIZE": per["FETCH_SIZE"][1], "WRITE_SIZE": per["WRITE_SIZE"][1]}, "seconds": time.perf_counter() - t0}


def measure_traffic_live(a, timeout_s: float = 240.0):
    """roofline.traffic of THIS run: measure_kernel_traffic over a one-step child run of this very command (same pairs per
    step = same images per launch) for the dominant kernel."""
    child = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--pairs", str(a.pairs), "--no-cpu-baseline", "--main-region-only",
             "--no-hook-path", "--no-strong-scaling", "--no-live-traffic", "--no-adaptive"]
    if a.lib:
        child += ["--lib", a.lib]
    return measure_kernel_traffic(child, "conv3x3_x6_kernel<64, 1, 1, true, 2", timeout_s)


def strong_scaling_subrun(a, rank, world, dev, dist, lib, line):
    """Strong scaling where the driver's 1/2/4/8 command sees it: the FIXED config-4 job (150 images -> 10 000 exhaustive
    pairs, images and pairs sharded over the ranks, two all-gathers) after the headline region; `value` stays the headline.  ``line`` (rank 0's
    headline record, complete before this starts) receives the `strong_scaling` sub-record.  Failure handling, exercised at world 3 over gloo by
    tests/test_bench_multirank_gloo.py: an exception on every rank becomes `strong_scaling.error`; a failure on ONE rank only (OOM, a HIP error)
    leaves the others inside an all-gather.  Every rank therefore runs a watchdog thread that ends the process when the sub-run has
    not returned after --strong-timeout seconds OR as soon as a peer reports a failure through the rendezvous store — rank 0 printing the
    headline line first; the failing rank waits for rank 0's acknowledgement (a launcher kills the surviving ranks the moment one exits
    non-zero) and then leaves without entering another collective."""
    store = None
    if dist is not None:
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
    done = threading.Event()

    def leave(reason):
        if rank == 0:
            line["strong_scaling"] = {"error": reason}
            line["cpu_baseline"] = None
            print(json.dumps(line), flush=True)
            if store is not None:
                try:
                    store.set("dim_strong_ack", "1")
                except Exception:
                    pass
        os._exit(3)

    def watch():
        t0 = time.perf_counter()
        limit = a.strong_timeout + (0.0 if rank == 0 else 5.0)
        while not done.wait(0.5):
            if time.perf_counter() - t0 > limit:
                leave(f"the config-4 sub-run did not return within {a.strong_timeout} s")
            if store is not None:
                try:
                    if store.check(["dim_strong_fail"]):
                        leave("a peer rank failed inside the config-4 sub-run: " + store.get("dim_strong_fail").decode(errors="replace")[:300])
                except Exception:
                    pass

    watchdog = threading.Thread(target=watch, daemon=True)
    watchdog.start()
    strong = None
    try:
        rec = measure_config4(a, rank, world, dev, dist, lib, steps=1, warmup=1)
        if rec is not None:
            strong = {k: rec[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "phases_s_max_over_ranks", "matches_total",
                                          "matches_per_pair_mean", "pairs_with_at_least_100_matches", "fp16x3_range_guard", "gathered_bytes_measured")}
            strong["workload"] = rec["config"]["workload"]
            strong["sharding"] = rec["config"]["sharding"]
    except Exception as e:
        strong = {"error": repr(e)[:400]}
        if world > 1:    # the peers may be blocked in a collective this rank will never join: report, wait for rank 0's line, leave
            if rank == 0:
                leave(strong["error"])
            if store is not None:
                try:
                    store.set("dim_strong_fail", f"rank {rank}: {strong['error']}")
                    store.wait(["dim_strong_ack"], __import__("datetime").timedelta(seconds=15))
                except Exception:
                    pass
            os._exit(3)
    done.set()
    if rank == 0:
        line["strong_scaling"] = strong


def main():
    a = parse()
    torch.set_num_threads(min(16, os.cpu_count() or 16))     # the GPU box has 256 host CPUs: a 256-thread intra-op pool only adds spin-waiting
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    if a.workload == "config1" and a.cpu_only:
        return run_config1(a, None, None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    sp = importlib.import_module(PKG + ".superpoint_hip")
    lg = importlib.import_module(PKG + ".lightglue_hip")
    weights = importlib.import_module(PKG + ".weights")
    capi = importlib.import_module(PKG + ".capi")
    if a.lib:  # developer knob: A/B a differently built library in the same GPU call
        capi.install(capi.load(a.lib), None)
    lib = capi.load()
    for kv in a.tune:
        k, v = kv.split("=")
        lib.dim_tune_set(int(k), int(v))

    if a.workload == "config4":
        return run_config4(a, rank, world, dev, dist, lib)
    if a.workload == "config5":
        return run_config5(a, rank, world, dev, dist, lib)
    if a.workload == "config1":
        assert world == 1, "config 1 is a 5-image job: one GPU"
        return run_config1(a, dev, lib)
    P, K, W = a.pairs, a.steps, a.warmup
    cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
    conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
    ext = sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234), cfg, max_batch=2 * P, max_hw=(1024, 1024),
                           capacity=2048, device=dev)
    mat = lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0, 256), conf, max_pairs=P, max_kpts=2048, device=dev)
    NK = mat.nk

    # synthetic pairs resident in HBM before the timed region: a pool of distinct batches
    n_pool = min(K + W, 4)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    pool = [torch.rand(2 * P, 1024, 1024, generator=g).to(dev) for _ in range(n_pool)]
    size_tab = torch.full((2 * P, 2), 1024.0, device=dev)
    # per-rank match table of the whole job in exchange format (SURVEY 8(e) phase 4): ONE flat int32 buffer
    # [counts: T*P | rows: T*P*NK x (idx0, idx1, score bits)], filled step by step (dim_op_pack_match_rows: 12-byte rows, zero
    # beyond a pair's count), exchanged with ONE all-gather at the end.  dim_lg_match's own outputs are double-buffered.
    T = K
    flat = torch.zeros(T * P + T * P * NK * 3, dtype=torch.int32, device=dev)
    cnt_tab, row_tab = flat[: T * P].view(T, P), flat[T * P:].view(T, P, NK, 3)
    outs = [{
        "matches": torch.zeros(P, NK, 2, dtype=torch.int64, device=dev),
        "scores": torch.zeros(P, NK, dtype=torch.float32, device=dev),
        "n_matches": torch.zeros(P, dtype=torch.int32, device=dev),
        "matches01": torch.zeros(P, 2, NK, dtype=torch.int32, device=dev),
        "mscores01": torch.zeros(P, 2, NK, dtype=torch.float32, device=dev),
        "stop": torch.zeros(P, dtype=torch.int32, device=dev),
        "prune01": torch.zeros(P, 2, NK, dtype=torch.int32, device=dev),
    } for _ in range(2)]

    # Main timed region: extraction and matching back-to-back on one stream, so that a kernel's HIP-event
    # duration is its own (the roofline figure) and agrees with the rocprofv3 trace.  The two-stream schedule —
    # while LightGlue matches batch i on stream B, SuperPoint already extracts batch i+1 on stream A
    # (double-buffered feature tables; power-limited MFMA convolutions and latency-bound attention / GEMMs
    # share the CUs) — gives the same rate within 0.02 % since round 4's kernels fill the chip on their own (BENCH_r05: 605.22 vs 605.13
    # pairs/s); --other-schedule times it right after the main region over the same K steps, --overlap makes it the main region.
    overlap = bool(a.overlap)
    s0 = torch.cuda.current_stream(dev)
    sA2, sB2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    feats = [tuple(t.clone() for t in ext.extract_batch(pool[0])) for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    done_lg = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()

    def run(n_steps, first, two_streams):
        """n_steps pipelined steps starting at pool index `first`; match tables go to slots 0..n_steps-1."""
        sA, sB = (sA2, sB2) if two_streams else (s0, s0)
        with torch.cuda.stream(sA):
            ext.extract_batch(pool[first % n_pool], out=feats[0])
            ev[0].record(sA)
        for i in range(n_steps):
            cur, nxt = i % 2, (i + 1) % 2
            if i + 1 < n_steps:
                with torch.cuda.stream(sA):
                    if i >= 1:
                        sA.wait_event(done_lg[nxt])  # the matcher of step i-1 is done with that feature buffer
                    ext.extract_batch(pool[(first + i + 1) % n_pool], out=feats[nxt])
                    ev[nxt].record(sA)
            with torch.cuda.stream(sB):
                sB.wait_event(ev[cur])
                kp, sc, de, n = feats[cur]
                o = mat.match_batch(kp, de, n, size_tab, n_pairs=P, out=outs[cur])
                capi.check(lib, lib.dim_op_pack_match_rows(capi.ptr(o["matches"]), capi.ptr(o["scores"]), capi.ptr(o["n_matches"]), NK, P,
                                                            capi.ptr(row_tab[i % T]), ctypes.c_void_p(sB.cuda_stream)))
                cnt_tab[i % T].copy_(o["n_matches"])
                done_lg[cur].record(sB)
        torch.cuda.current_stream(dev).wait_stream(sA)
        torch.cuda.current_stream(dev).wait_stream(sB)
        return feats[(n_steps - 1) % 2][3]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if W > 0:
        run(W, 0, overlap)
    barrier()
    # time every launch of conv3x3_x6_kernel<64,1,1,true> (conv1a fused into conv1b) with HIP events on the launch stream
    capi.check(lib, lib.dim_profile_start(ctypes.c_ulonglong(1 << 1)))  # DIM_PROF_SP_CONV1B: the fused conv1a+conv1b kernel
    clk = torch.zeros(4, dtype=torch.int64, device=dev)  # {shader cycles, 100 MHz ticks} before / after the timed region
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    capi.check(lib, lib.dim_op_read_clocks(ctypes.c_void_p(clk.data_ptr()), stream_ptr))
    t0 = time.perf_counter()
    n_last = run(K, W, overlap)
    capi.check(lib, lib.dim_op_read_clocks(ctypes.c_void_p(clk.data_ptr() + 16), stream_ptr))
    if dist is not None:  # ONE collective for the whole job: per-rank match tables (counts + 12-byte rows) -> every rank
        flat_all = torch.empty(world * flat.numel(), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(flat_all, flat)
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches = ctypes.c_double(), ctypes.c_int()
    capi.check(lib, lib.dim_profile_stop(ctypes.byref(tot_ms), ctypes.byref(launches)))

    # the same kernel timed again with nothing else on the GPU (2 extraction-only batches after the timed region)
    iso_ms, iso_n = ctypes.c_double(), ctypes.c_int()
    capi.check(lib, lib.dim_profile_start(ctypes.c_ulonglong(1 << 1)))
    for i in range(2):
        ext.extract_batch(pool[i % n_pool], out=feats[0])
    torch.cuda.synchronize()
    capi.check(lib, lib.dim_profile_stop(ctypes.byref(iso_ms), ctypes.byref(iso_n)))

    # the other schedule over the same K steps, timed the same way (not part of `value`)
    dt2 = float("nan")
    if a.other_schedule and not a.main_region_only:
        run(1, 0, not overlap)
        barrier()
        t1 = time.perf_counter()
        run(K, W, not overlap)
        barrier()
        dt2 = time.perf_counter() - t1

    tmax = torch.tensor([dt, dt2], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, dt2 = float(tmax[0].item()), float(tmax[1].item())
    n_kpts_ok = bool((n_last == 2048).all().item())
    ck = clk.cpu().tolist()
    clock_mhz = (ck[2] - ck[0]) / max(1, ck[3] - ck[1]) * 100.0
    sat_total, sat_sites = capi.saturation(lib, stream_ptr, reset=True)  # fp16x3 range guard over the whole run: must be 0

    flat_numel = flat.numel()
    line = None
    if rank == 0:
        pairs_total = world * K * P
        value = pairs_total / dt
        conv_ms = tot_ms.value / max(1, launches.value)
        gflop_per_launch = (CONV1A_GFLOP_PER_IMAGE + CONV1B_GFLOP_PER_IMAGE) * 2 * P  # one launch = the 2P images of a step
        conv_tflops = gflop_per_launch / conv_ms  # GFLOP / ms == TFLOP/s
        traffic = None
        pmc = ROOT / "profiles" / "conv1b_hbm_bytes.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())  # measured at rec["batch_images_per_launch"] images per launch; scales with the batch
                traffic = rec.get("hbm_bytes_per_launch") * (2 * P) / rec.get("batch_images_per_launch", 2 * P)
            except Exception:
                traffic = None
        line = {
            "metric": "image-pairs/s (SuperPoint+LightGlue, 1024^2, 2048 kpts)",
            "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "arithmetic": "fp32 results; matrix products as 2-way fp16 splits of power-of-two-scaled operands x 3 fp16-MFMA cross terms (fp32-class accuracy, measured <= the fp32 fmaf chain's error)", "data": "synthetic",
            "config": {"workload": "configs[2]: SuperPoint+LightGlue, synthetic 1024x1024 grayscale pairs @2048 kpts, "
                                   "2 extractions + 1 match per pair, LightGlue fixed-work (9 layers, no early stop/pruning), "
                                   "seeded synthetic weights", "pairs_per_step_per_gpu": P, "image": "1024x1024",
                       "keypoints": 2048, "all_2048_kpts": n_kpts_ok, "gflop_per_pair": 2 * SP_GFLOP_PER_IMAGE + LG_GFLOP_PER_PAIR,
                       "sharding": f"pairs sharded over {world} rank(s); ONE RCCL all-gather of the match tables at the end "
                                   f"(flat int32: counts + (idx0, idx1, score) rows, {flat_numel * 4 / 1e6:.1f} MB per rank)",
                       "streams": "extraction of batch i+1 overlaps matching of batch i (2 HIP streams)" if overlap else "single stream"},
            ("single_stream" if overlap else "two_stream_overlap"): None if (a.main_region_only or not a.other_schedule) else {
                "value": pairs_total / dt2, "unit": "image-pairs/s", "ms_per_step": dt2 / K * 1e3,
                "note": "the same K steps under the other schedule, timed right after the main region (barrier + synchronize on both sides)"},
            "end_to_end_tflops_per_gpu": (2 * SP_GFLOP_PER_IMAGE + LG_GFLOP_PER_PAIR) * value / world / 1e3,
            "timed_region_s": dt, "pairs_total": pairs_total, "sustained_clock_mhz": clock_mhz,
            "fp16x3_range_guard": {"violations": sat_total, "sites": sat_sites},
            "strong_scaling": None,
            "hook_path": None,
            "adaptive": None,
            "roofline": {"kernel": "conv3x3_x6_kernel<64,1,1,true,2> (SuperPoint conv1a 1->64 evaluated in the halo staging + conv1b 64->64 3x3 "
                                   "+ bias + ReLU + 2x2 max-pool, fp32-accurate on the fp16 MFMA (fp16x3); 1024^2 images)", "bound": "mfma",
                         "achieved": conv_tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": conv_tflops / PEAK_BF16_MFMA_TFLOPS,
                         "mfma_pipe_util": conv_tflops * X6_PASSES / PEAK_BF16_MFMA_TFLOPS,
                         "peak_note": "frac = ALGORITHMIC fp32 FLOP/s / dense fp16 MFMA peak (2500 TFLOP/s, the precision issued; the 3 "
                                      "split-precision passes are not credited, so frac <= 1/3); mfma_pipe_util = x 3 passes = what the "
                                      "MFMA-busy counter shows (profiles/r03_pmc_mfma_summary.txt)",
                         "frac_of_fp32_mfma_peak": conv_tflops / PEAK_FP32_MFMA_TFLOPS,
                         "power_limited": {"sustained_peak": SUSTAINED_FP16_MFMA_TFLOPS, "mfma_issue_rate": conv_tflops * X6_PASSES,
                                           "frac_of_sustained": conv_tflops * X6_PASSES / SUSTAINED_FP16_MFMA_TFLOPS,
                                           "note": "CITED, not measured by this run: the rate back-to-back fp16 MFMAs sustain on every SIMD with operands "
                                                   "like this kernel's, held at 1.65 GHz by the 1314 W package power limit "
                                                   "(profiles/r04_mfma_power_probe.jsonl; 2450 TFLOP/s only with all-zero operands); mfma_issue_rate = "
                                                   "achieved x 3 split passes"},
                         "traffic": traffic,
                         "traffic_note": "CITED, not measured by this run: HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                         "of the same kernel (profiles/conv1b_hbm_bytes.json names the pass), scaled to this run's images per launch",
                         "isolated": {"note": "same kernel, same launches, timed again right after the timed region in two extraction-only batches "
                                              "(the main region is single-stream, so the two agree unless --overlap is given)",
                                      "avg_launch_ms": iso_ms.value / max(1, iso_n.value),
                                      "achieved": gflop_per_launch / (iso_ms.value / max(1, iso_n.value)),
                                      "frac": gflop_per_launch / (iso_ms.value / max(1, iso_n.value)) / PEAK_BF16_MFMA_TFLOPS},
                         "avg_launch_ms": conv_ms, "launches": launches.value,
                         "algorithmic_gflop_per_launch": gflop_per_launch,
                         # what the launch must move: the fp32 image in, the pooled 64-channel map out as two fp16 planes (= 4 bytes per value)
                         "algorithmic_hbm_bytes_per_launch": (1024 * 1024 * 4 + 512 * 512 * 64 * 4) * 2 * P},
        }
    # the per-call plugin hooks at the headline sizes: measured before the strong-scaling sub-run, whose host-side image
    # synthesis spins up torch's CPU thread pool
    if rank == 0 and not a.no_hook_path and not a.main_region_only:
        try:
            line["hook_path"] = measure_hook_path(dev, lib)
        except Exception as e:
            line["hook_path"] = {"error": repr(e)[:400]}
    if rank == 0 and not a.no_adaptive and not a.main_region_only:
        try:
            line["adaptive"] = measure_adaptive(dev, lib)
        except Exception as e:
            line["adaptive"] = {"error": repr(e)[:400]}
    if not a.no_strong_scaling and not a.main_region_only:
        del pool, feats, flat, outs
        torch.cuda.empty_cache()
        strong_scaling_subrun(a, rank, world, dev, dist, lib, line)
    if rank == 0 and world == 1 and not a.no_live_traffic and not a.main_region_only:
        # after everything that is timed: two counter passes over a one-step child run (the GPU is idle, this process's batch tables are released)
        try:
            tr = measure_traffic_live(a)
            line["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            line["roofline"]["traffic_note"] = ("MEASURED by this run: HBM bytes per launch of this kernel from two separate rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE / "
                                                "--pmc WRITE_SIZE) over a one-step child run of the same command, FETCH_SIZE doubled (MI355X_MICROARCH.md section HBM)")
            line["roofline"]["traffic_measurement"] = tr
        except Exception as e:
            line["roofline"]["traffic_measurement"] = {"error": repr(e)[:300], "fallback": "the cited value (traffic_note)"}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.cpu_sample_pairs)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
