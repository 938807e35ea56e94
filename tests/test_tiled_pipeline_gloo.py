"""CPU, world_size 2 over gloo: BASELINE config 5's multi-GPU path (pipeline.TiledPairPipeline: tile-wise ALIKED extraction of the images
i mod world -> ONE all-gather of the merged tile tables -> tile selection of the image pairs j mod world -> ONE small all-gather of the
selection masks -> cost-balanced deal of the image pairs, tile pairs matched as one batched stream -> ONE all-gather of the match rows)
must give every rank exactly the single-process result, and the single-process result must equal the per-pair numpy path
(tile_matching.match_tile_pairs_batched = the reference's loop MB:414-460).  Device work = the HIP sources on the test emulator, through the
plugin classes (capi.install hook), RGB images, 128-d descriptors."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _run(lib_path, rank, world):
    import ctypes

    sys.path.insert(0, str(ROOT))
    capi = importlib.import_module("deep-image-matching_amd.capi")
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    pl = importlib.import_module("deep-image-matching_amd.pipeline")
    lib = ctypes.CDLL(lib_path)
    capi.install(lib, "cpu")
    general = {"tile_size": (96, 64), "tile_overlap": 0, "min_matches_per_tile": 1, "quality": "HIGH"}
    ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 24,
                                                                     "detection_threshold": 0.2, "nms_radius": 2, "allow_synthetic_weights": True}})
    mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1,
                                                                   "filter_threshold": 0.0, "allow_synthetic_weights": True}}, local_features="aliked")
    rng = np.random.default_rng(21)
    base = (rng.random((128 + 32, 192 + 32, 3)) * 255).astype(np.float32)
    images = [np.ascontiguousarray(base[dy:dy + 128, dx:dx + 192]) for dy, dx in ((0, 0), (32, 0), (0, 32))]      # 2 x 2 tiles of 96 x 64 each
    pipe = pl.TiledPairPipeline(ex, mt, rank, world, selection="GRID")
    feats = pipe.extract_all(images)
    pairs = pl.exhaustive_pairs(3)
    matches = pipe.match_all(images, feats, pairs)
    if world == 1:      # independent of the pipeline's batched stream: the per-pair numpy path on the same features and GRID tile pairs
        tm = importlib.import_module("deep-image-matching_amd.tile_matching")
        grid = tm.select_tile_pairs("GRID", range(4), range(4))
        for (a, b), m in zip(pairs.tolist(), matches):
            want = tm.match_tile_pairs_batched(mt._ensure_pairs, feats[a], feats[b], grid, "cpu", 3)
            assert np.array_equal(m, want), (a, b, m.shape, want.shape)
        assert pipe.timings["tile_pairs_total"] == 12
    return feats, matches


def _worker(rank, world, port, lib_path, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = []
    orig = dist.all_gather_into_tensor

    def counting(out, inp, *a, **k):
        calls.append((inp.dtype, inp.numel()))
        return orig(out, inp, *a, **k)

    dist.all_gather_into_tensor = counting
    feats, matches = _run(lib_path, rank, world)
    dist.all_gather_into_tensor = orig
    torch.save({"feats": feats, "matches": matches, "collectives": calls}, os.path.join(out_dir, f"tiled{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tiled_pipeline_equals_single_process(tmp_path):
    build = importlib.import_module("deep-image-matching_amd.build")
    lib_path = str(build.build_emu())
    capi = importlib.import_module("deep-image-matching_amd.capi")
    try:
        feats1, matches1 = _run(lib_path, 0, 1)
    finally:
        capi.install(None)
    assert all(f["keypoints"].shape[0] > 0 and f["descriptors"].shape[0] == 128 for f in feats1)
    assert {int(t) for f in feats1 for t in np.unique(f["tile_idx"])} == {0, 1, 2, 3}
    assert sum(m.shape[0] for m in matches1) > 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, lib_path, str(tmp_path)), nprocs=2, join=True)
    cap = 4 * 24                                     # 4 tiles x 24 keypoints: the exchange slot of one image
    for r in range(2):
        got = torch.load(tmp_path / f"tiled{r}.pt", weights_only=False)
        for a, b in zip(got["feats"], feats1):
            assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
        assert len(got["matches"]) == 3 and all(np.array_equal(a, b) and a.dtype == np.int64 for a, b in zip(got["matches"], matches1))
        # phase 2: one fp32 buffer of 2 image slots x cap x (2 + 1 + 1 + 128) + 2 counts; phase 3a: the selection masks, 2 pair slots x (4 x 4 tiles)
        # bytes; phase 4: one int32 buffer of 2 pair slots x (count + bound rows x 2), bound = max over the pairs of sum(min(n0, n1)) over the
        # selected tile pairs — the exact upper bound every rank derives from the masks and the tile counts
        cnt = [np.bincount(f["tile_idx"].astype(np.int64), minlength=4) for f in feats1]
        bound = max(sum(min(int(cnt[a][t]), int(cnt[b][t])) for t in range(4)) for a, b in ((0, 1), (0, 2), (1, 2)))
        assert got["collectives"] == [(torch.float32, 2 * cap * 132 + 2), (torch.uint8, 2 * 16), (torch.int32, 2 + 2 * bound * 2)], got["collectives"]


def test_device_resident_tile_matching_equals_the_numpy_path():
    """pipeline.TiledPairPipeline keeps the merged tile tables in HBM and matches from there (tile_matching.match_tile_pairs_batched_device);
    the array it returns must be the one the numpy path (match_tile_pairs_batched = the reference's loop MB:414-460, batched) returns."""
    import ctypes
    build = importlib.import_module("deep-image-matching_amd.build")
    capi = importlib.import_module("deep-image-matching_amd.capi")
    tm = importlib.import_module("deep-image-matching_amd.tile_matching")
    pl = importlib.import_module("deep-image-matching_amd.pipeline")
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    lib = ctypes.CDLL(str(build.build_emu()))
    capi.install(lib, "cpu")
    try:
        general = {"tile_size": (96, 64), "tile_overlap": 0, "min_matches_per_tile": 1, "quality": "HIGH"}
        ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 24,
                                                                         "detection_threshold": 0.2, "nms_radius": 2, "allow_synthetic_weights": True}})
        mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1,
                                                                       "filter_threshold": 0.0, "allow_synthetic_weights": True}}, local_features="aliked")
        rng = np.random.default_rng(22)
        base = (rng.random((128 + 32, 192 + 32, 3)) * 255).astype(np.float32)
        images = [np.ascontiguousarray(base[dy:dy + 128, dx:dx + 192]) for dy, dx in ((0, 0), (32, 32))]
        pipe = pl.TiledPairPipeline(ex, mt, 0, 1, selection="GRID")
        feats = pipe.extract_all(images)                       # numpy dicts; pipe._dev_feats = the same tables in "device" memory
        # the device-returning merge equals the numpy-returning one
        direct = ex._extract_by_tile(images[0])
        assert all(np.array_equal(direct[k], feats[0][k]) for k in ("keypoints", "descriptors", "scores", "tile_idx"))
        pairs = tm.select_tile_pairs("GRID", range(4), range(4)) + [(0, 3), (2, 1), (1, 1)]
        want = tm.match_tile_pairs_batched(mt._ensure_pairs, feats[0], feats[1], pairs, "cpu", 3)
        got = tm.match_tile_pairs_batched_device(mt._ensure_pairs, pipe._dev_feats[0], pipe._dev_feats[1], pairs, 3)
        assert want.shape[0] > 0 and got.dtype == torch.int64 and np.array_equal(got.numpy(), want)
        # the cached device tables serve only the list extract_all returned; a modified copy is uploaded, never silently replaced
        flt = [dict(f) for f in feats]
        keep = flt[0]["tile_idx"] != 3
        flt[0] = {k: (v[keep] if k in ("keypoints", "scores", "tile_idx") else (v[:, keep] if k == "descriptors" else v)) for k, v in flt[0].items()}
        m_all = pipe.match_all(images, feats, pl.exhaustive_pairs(2))[0]
        m_flt = pipe.match_all(images, flt, pl.exhaustive_pairs(2))[0]
        grid = tm.select_tile_pairs("GRID", range(4), range(4))
        assert np.array_equal(m_all, tm.match_tile_pairs_batched(mt._ensure_pairs, feats[0], feats[1], grid, "cpu", 3))
        assert np.array_equal(m_flt, tm.match_tile_pairs_batched(mt._ensure_pairs, flt[0], flt[1], grid, "cpu", 3)) and not np.array_equal(m_all, m_flt)
        none = tm.match_tile_pairs_batched_device(mt._ensure_pairs, pipe._dev_feats[0], pipe._dev_feats[1], [], 3)
        assert none.shape == (0, 2)
    finally:
        capi.install(None)


def _run_presel(lib_path, rank, world):
    """three crops of one scene (shifts at multiples of 8 px), PRESELECTION with matching-capable preselector weights; returns (matches, n_fallback,
    tile_pairs_total).  pipeline._band1 (the host-side first-band extraction) is replaced by a function that fails: the selection phase must live on the
    preselection features that travelled with the tile tables."""
    import ctypes

    sys.path.insert(0, str(ROOT))
    capi = importlib.import_module("deep-image-matching_amd.capi")
    plugins = importlib.import_module("deep-image-matching_amd.plugins")
    pl = importlib.import_module("deep-image-matching_amd.pipeline")
    tm = importlib.import_module("deep-image-matching_amd.tile_matching")
    weights = importlib.import_module("deep-image-matching_amd.weights")
    lib = ctypes.CDLL(lib_path)
    capi.install(lib, "cpu")
    old_kp, old_band = tm.PRESELECTION_SP_CONF["max_keypoints"], pl._band1
    tm.PRESELECTION_SP_CONF["max_keypoints"] = 256
    try:
        general = {"tile_size": (96, 64), "tile_overlap": 0, "min_matches_per_tile": 1, "quality": "HIGH", "tile_preselection_size": 192,
                   "allow_synthetic_weights": True}
        ex = plugins.AlikedExtractor({"general": general, "extractor": {"name": "aliked", "model_name": "aliked-n16rot", "max_num_keypoints": 24,
                                                                         "detection_threshold": 0.2, "nms_radius": 2, "allow_synthetic_weights": True}})
        mt = plugins.LightGlueMatcher({"general": general, "matcher": {"name": "lightglue", "n_layers": 2, "depth_confidence": -1, "width_confidence": -1,
                                                                       "filter_threshold": 0.0, "allow_synthetic_weights": True}}, local_features="aliked")
        rng = np.random.default_rng(23)
        base = (rng.random((128 + 64, 192 + 64, 3)) * 255).astype(np.float32)
        images = [np.ascontiguousarray(base[dy:dy + 128, dx:dx + 192]) for dy, dx in ((0, 0), (32, 48), (16, 24))]
        sp_sd = weights.synthetic_superpoint_state_dict(1234)
        pre = tm.TilePreselector(sp_sd, weights.synthetic_lightglue_state_dict(0, 256), tile_preselection_size=192, device="cpu", lib=lib)
        f0 = pre.features("warm", np.ascontiguousarray(images[0][..., 0]), "HIGH")
        center = f0[1][0, : int(f0[2][0])].mean(0)
        mt._tile_preselector = tm.TilePreselector(sp_sd, weights.synthetic_lightglue_matching_state_dict(0, 256, center=center), tile_preselection_size=192,
                                                  device="cpu", lib=lib)
        pipe = pl.TiledPairPipeline(ex, mt, rank, world, selection="PRESELECTION")
        names = ["a", "b", "c"]
        feats = pipe.extract_all(images, names=names)

        def no_pixels(image):
            raise AssertionError("the selection phase read an image's pixels")

        pl._band1 = no_pixels
        pairs = pl.exhaustive_pairs(3)
        matches = pipe.match_all(images, feats, pairs, names=names)
        if world == 1:
            # the pipeline's batched selection (all image pairs' preselector LightGlue calls in one stream, vote kernels, one read-back) must select
            # exactly what the per-pair host path (BatchedTileMatchingMixin.tile_selection) selects — and must actually VOTE
            for (a, b), m in list(zip(pairs.tolist(), matches))[:1]:      # (one image pair: every further one costs the emulator ~20 s)
                sel = mt.tile_selection(names[a], names[b], "PRESELECTION", image0=np.ascontiguousarray(images[a][..., 0]), image1=np.ascontiguousarray(images[b][..., 0]))
                assert len(sel) > 0
                want = tm.match_tile_pairs_batched(mt._ensure_pairs, feats[a], feats[b], sel, "cpu", 3)
                assert np.array_equal(m, want), (a, b, sel)
        return matches, pipe.n_fallback, pipe.timings["tile_pairs_total"], (4 * 24, mt._preselector()._capacity())      # (exchange slot: 4 tiles x 24 keypoints)
    finally:
        tm.PRESELECTION_SP_CONF["max_keypoints"] = old_kp
        pl._band1 = old_band



def _presel_worker(rank, world, port, lib_path, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    calls = []
    orig = dist.all_gather_into_tensor

    def counting(out, inp, *a, **k):
        calls.append((inp.dtype, inp.numel()))
        return orig(out, inp, *a, **k)

    dist.all_gather_into_tensor = counting
    matches, n_fb, n_tp, caps = _run_presel(lib_path, rank, world)
    dist.all_gather_into_tensor = orig
    torch.save({"matches": matches, "n_fallback": n_fb, "tile_pairs": n_tp, "collectives": calls, "caps": caps}, os.path.join(out_dir, f"presel{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_preselection_travels_with_the_tile_tables(tmp_path):
    """PRESELECTION through the pipeline, one and two ranks.  One process: the batched selection equals the per-pair host path and votes for real tile
    pairs (matching-capable preselector weights on crops whose down-sampled overlaps are shifted copies; no fallback configured, none needed).  Two ranks
    (round 5): the down-sampled SuperPoint features of every image ride in the feature all-gather (a third section of the exchange buffer), so the
    selection phase of EITHER rank never reads pixels — also not for image pairs whose images the other rank extracted — and gives the single-process
    result; still exactly three collectives."""
    build = importlib.import_module("deep-image-matching_amd.build")
    lib_path = str(build.build_emu())
    capi = importlib.import_module("deep-image-matching_amd.capi")
    try:
        matches1, n_fb1, n_tp1, _ = _run_presel(lib_path, 0, 1)
    finally:
        capi.install(None)
    assert n_fb1 == 0 and n_tp1 > 0 and sum(m.shape[0] for m in matches1) > 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_presel_worker, args=(2, port, lib_path, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(tmp_path / f"presel{r}.pt", weights_only=False)
        assert len(got["matches"]) == 3 and all(np.array_equal(a, b) for a, b in zip(got["matches"], matches1)), r
        cap, pcap = got["caps"]
        per = 2                                    # 3 images over 2 ranks
        feat = per * cap * 132 + per + per * pcap * 258 + 2 * per
        assert len(got["collectives"]) == 3 and got["collectives"][0] == (torch.float32, feat) and got["collectives"][1] == (torch.uint8, 2 * 16), got["collectives"]
