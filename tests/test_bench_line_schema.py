"""The committed bench line (profiles/r05_bench_n1.json = `python bench.py` on one MI355X) carries the contract's fields and is
self-consistent: metric / unit are BASELINE.json's, value = pairs / timed region, roofline.frac = achieved / peak with the algorithmic work
stated, the cpu_baseline object is complete, the strong_scaling sub-record ran the fixed config-4 job with non-empty match tables."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _line():
    return json.loads((ROOT / "profiles" / "r05_bench_n1.json").read_text().strip().splitlines()[-1])


def test_contract_fields_and_baseline_metric():
    d, base = _line(), json.loads((ROOT / "BASELINE.json").read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "image-pairs/s" and "image-pairs/s" in base["metric"] and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - d["pairs_total"] / d["timed_region_s"]) < 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - d["timed_region_s"] / d["steps"] * 1e3) < 1e-6 * d["ms_per_step"]
    assert d["fp16x3_range_guard"]["violations"] == 0 and d["config"]["all_2048_kpts"] is True


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.1 < r["frac"] <= 1.0 / 3.0          # three fp16 passes per fp32-accurate product
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_ms"]) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    tm = r.get("traffic_measurement")
    if tm is not None and "error" not in tm:   # measured by the run itself (two rocprofv3 --pmc passes over a one-step child run): within a few % of the algorithmic bytes
        assert r["traffic"] == tm["hbm_bytes_per_launch"] and "MEASURED" in r["traffic_note"]
        assert 0.95 < r["traffic"] / r["algorithmic_hbm_bytes_per_launch"] < 1.15 and min(tm["launches_counted"].values()) >= 2
    p = r["power_limited"]
    assert abs(p["frac_of_sustained"] - 3 * r["achieved"] / p["sustained_peak"]) < 1e-9 and p["frac_of_sustained"] < 1.0
    c = d["cpu_baseline"]
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert d["value"] / c["value"] > 100            # (a sanity bound, not a claim: the roofline fraction is the quality figure)


def test_strong_scaling_sub_record_is_the_fixed_job_with_real_matches():
    s = _line()["strong_scaling"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 1 and s["value"] > 0
    assert s["matches_per_pair_mean"] >= 100 and s["pairs_with_at_least_100_matches"] > 0 and s["fp16x3_range_guard"]["violations"] == 0
    assert set(s["phases_s_max_over_ranks"]) == {"extract_s", "feature_gather_s", "match_s", "match_gather_s"}


def test_hook_path_sub_record_and_config1_line():
    """Round 5: the default line carries the per-call plugin hooks' wall time beside the device time of the same batch-1 calls,
    and profiles/r05_config1.json is BASELINE configs[0] on its real inputs with the reference's CPU path beside it (next #1)."""
    h = _line()["hook_path"]
    assert set(h) >= {"ms_per_image", "ms_per_pair", "device_only_ms_per_image", "device_only_ms_per_pair"}
    assert h["device_only_ms_per_image"] <= h["ms_per_image"] < 3.0 and h["device_only_ms_per_pair"] <= h["ms_per_pair"] < 4.0
    c = json.loads((ROOT / "profiles" / "r05_config1.json").read_text().strip().splitlines()[-1])
    assert "configs[0]" in c["config"]["workload"] and c["config"]["images"] == 5 and c["config"]["pairs"] == 10 and c["n_gpus"] == 1
    assert abs(c["value"] - 10 / (c["hook_path"]["extract_s"] + c["hook_path"]["match_s"])) < 1e-6 * c["value"]
    assert c["fp16x3_range_guard"]["violations"] == 0 and c["batched_image_matcher"]["value"] > 0
    assert c["cpu_baseline"]["kind"] in ("port", "reference") and c["value"] / c["cpu_baseline"]["value"] > 50
    ref = json.loads((ROOT / "profiles" / "r05_config1_cpu_reference.json").read_text().strip().splitlines()[-1])["cpu_baseline"]
    assert ref["kind"] == "reference" and 0.5 < ref["value"] < 5.0 and "imported from /root/reference" in ref["sample"]
