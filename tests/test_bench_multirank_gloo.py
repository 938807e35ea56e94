"""CPU, world_size 8 (and 3) over gloo: the code the driver's `bench.py --gpus 8` will run for the first time on hardware — bench.measure_config4
(the `strong_scaling` sub-record / `--workload config4`: phases 1-4 of SURVEY 8(e)) and bench.strong_scaling_subrun (its watchdog / error
paths) — executed here at world 8 with uneven shards on the emulator build of the same HIP sources, scaled down (48 x 64 images, 32 keypoints);
and the per-rank buffer sizes of the FULL-size job at world 8 from the same formulas."""
import importlib
import json
import os
import socket
import subprocess
import sys
import types
from pathlib import Path

import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _args(images=11, job_pairs=16, pairs=2, timeout=900.0):      # (the watchdog ends the PROCESS: generous, the suite may run these on a loaded host)
    return types.SimpleNamespace(images=images, job_pairs=job_pairs, pairs=pairs, c4_hw=(48, 64), c4_kpts=32, strong_timeout=timeout)


def _setup(lib_path):
    import ctypes

    sys.path.insert(0, str(ROOT))
    capi = importlib.import_module("deep-image-matching_amd.capi")
    lib = ctypes.CDLL(lib_path)
    capi.install(lib, "cpu")
    return importlib.import_module("bench"), lib


def _worker(rank, world, port, lib_path, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    bench, lib = _setup(lib_path)
    calls = []
    orig = dist.all_gather_into_tensor

    def counting(out, inp, *a, **k):
        calls.append((str(inp.dtype), inp.numel()))
        return orig(out, inp, *a, **k)

    dist.all_gather_into_tensor = counting
    line = {"metric": "headline", "strong_scaling": None, "cpu_baseline": None}
    bench.strong_scaling_subrun(_args(), rank, world, torch.device("cpu"), dist, lib, line)
    dist.all_gather_into_tensor = orig
    torch.save({"line": line, "collectives": calls}, os.path.join(out_dir, f"bench{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_8_strong_scaling_subrecord_with_uneven_shards_equals_world_1(tmp_path):
    build = importlib.import_module("deep-image-matching_amd.build")
    lib_path = str(build.build_emu())
    capi = importlib.import_module("deep-image-matching_amd.capi")
    try:
        bench, lib = _setup(lib_path)
        one = {"strong_scaling": None}
        bench.strong_scaling_subrun(_args(), 0, 1, torch.device("cpu"), None, lib, one)
    finally:
        capi.install(None)
    s1 = one["strong_scaling"]
    assert "error" not in s1 and s1["n_gpus"] == 1 and s1["matches_total"] > 0, s1
    mp.spawn(_worker, args=(8, _free_port(), lib_path, str(tmp_path)), nprocs=8, join=True)
    got = [torch.load(tmp_path / f"bench{r}.pt", weights_only=False) for r in range(8)]
    s8 = got[0]["line"]["strong_scaling"]
    assert "error" not in s8 and s8["n_gpus"] == 8 and s8["scaling"] == "strong", s8
    # the same job, the same matches, whatever the sharding: 11 images over 8 ranks (2, 2, 2, 1, 1, 1, 1, 1), 16 pairs (2 per rank, dealt by cost)
    assert s8["matches_total"] == s1["matches_total"] and s8["pairs_with_at_least_100_matches"] == s1["pairs_with_at_least_100_matches"]
    assert set(s8["phases_s_max_over_ranks"]) == {"extract_s", "feature_gather_s", "match_s", "match_gather_s"}
    assert all(g["line"]["strong_scaling"] is None for g in got[1:])          # only rank 0 assembles the record
    # exactly two data collectives per pass (warm-up pass + timed pass), of the sizes the formulas give: per = ceil(n / world) slots
    per_i, per_p, cap, D, NK = 2, 2, 32, 256, 32
    feat, match = per_i * cap * (2 + 1 + D) + per_i, per_p * 2 + per_p * NK * 3
    big = [c for c in got[3]["collectives"] if c[1] in (feat, match)]
    assert ("torch.float32", feat) in big and ("torch.int32", match) in big, got[3]["collectives"]
    assert s8["gathered_bytes_measured"] == {"feature_gather_bytes": feat * 4 * 8, "match_gather_bytes": match * 4 * 8}
    assert bench.pipe_bytes(11, 8, cap, D, NK)[0] == feat * 4 * 8 and bench.pipe_bytes(16, 8, cap, D, NK)[1] == match * 4 * 8


def test_full_size_job_buffers_at_world_8_fit_the_gpu_many_times_over():
    """bench.pipe_bytes = the flat exchange buffers of pipeline.py; the full config-4 job (150 images, 10 000 pairs, 2048 keypoints, D 256) at 8 ranks."""
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    feat, match = bench.pipe_bytes(150, 8), bench.pipe_bytes(10000, 8)
    per_i, per_p = 19, 1250
    assert feat[0] == (per_i * 2048 * 259 + per_i) * 4 * 8 and match[1] == (per_p * 2 + per_p * 2048 * 3) * 4 * 8
    # per rank: its own slot + the gathered copy of all 8; plus the unpacked int64 / fp32 result tables every rank builds (P x NK x (16 + 4) bytes)
    per_rank = feat[0] * (1 + 1 / 8) + match[1] * (1 + 1 / 8) + 10000 * 2048 * 20
    assert per_rank < 1.2e9 and per_rank < 288e9 / 100


def _fail_worker(rank, world, port, lib_path, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DIM_BENCH_TEST_FAIL_RANK"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    bench, lib = _setup(lib_path)
    line = {"metric": "headline", "value": 1.0, "strong_scaling": None, "cpu_baseline": None}
    sys.stdout = open(os.path.join(out_dir, f"out{rank}.txt"), "w")
    bench.strong_scaling_subrun(_args(images=5, job_pairs=6, timeout=20.0), rank, world, torch.device("cpu"), dist, lib, line)
    print("RETURNED", flush=True)      # not reached on any rank: the failing rank exits, the others' collective fails or their watchdog fires


def test_failure_on_one_rank_only_ends_every_rank_and_rank_0_still_prints_the_headline_line(tmp_path):
    build = importlib.import_module("deep-image-matching_amd.build")
    lib_path = str(build.build_emu())
    code = ("import sys; sys.path.insert(0, %r); import torch.multiprocessing as mp; from tests.test_bench_multirank_gloo import _fail_worker, _free_port; "
            "mp.spawn(_fail_worker, args=(3, _free_port(), %r, %r), nprocs=3, join=True)" % (str(ROOT), lib_path, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert r.returncode != 0                                            # the ranks left with exit code 3: the launcher sees the failure
    out0 = (tmp_path / "out0.txt").read_text()
    assert "RETURNED" not in out0
    line = json.loads(out0.strip().splitlines()[-1])
    assert line["value"] == 1.0 and "error" in line["strong_scaling"], line   # the completed headline record, with the sub-run's failure recorded
    for rnk in (1, 2):
        assert "RETURNED" not in (tmp_path / f"out{rnk}.txt").read_text()
