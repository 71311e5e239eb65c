"""The JSON line `bench.py` prints is the driver's contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
scaling / vs_baseline / dtype / data / config.workload + roofline + cpu_baseline).  Checked here on the newest committed line of a real run
(profiles/*default_bench*.json, written on the GPU box by `python bench.py`), and on the pure helpers of bench.py - no GPU needed."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_newest_committed_default_line_has_the_contract_keys():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*default_bench*.json")))
    assert files, "no committed bench line under profiles/"
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "beam=5" in d["metric"] and "batch=100" in d["metric"] and d["vs_baseline"] is None          # BASELINE.md holds no published number
    assert abs(d["value"] - 100 * 20 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-3 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    if files[-1].split(os.sep)[-1] >= "r05":        # since round 5 the driver-kept `config` carries the other half of the metric too
        for k in ("xe_samples_per_s", "xe_ms_per_step", "xe_roofline_frac", "f32_exact_tokens_per_s", "greedy_tokens_per_s", "batch13_ms", "vs_cpu_baseline"):
            assert k in d["config"], k
        ratio = d["value"] / c["value"]
        if files[-1].split(os.sep)[-1] >= "r06_y":    # since round 6 the ratio is printed with 2 significant digits (the CPU leg moves +-10 % box to box)
            assert abs(d["config"]["vs_cpu_baseline"] - ratio) <= 0.05 * ratio and d["config"]["vs_cpu_baseline"] == float("%.2g" % ratio)
            assert d["config"]["cpu_baseline_timed_calls"] == c["timed_calls"]
        else:
            assert abs(d["config"]["vs_cpu_baseline"] - ratio) < 1e-6 * d["config"]["vs_cpu_baseline"]


def test_roofline_block_arithmetic():
    b = _bench_module()
    # 100 timed launches of 10 GFLOP each in 5 ms of GEMM time: 200 TFLOP/s against the three-MFMA fp32-equivalent peak
    r = b.roofline_block("f16x2", gemm_ms=5.0, gemm_n=100, gemm_seen=500, gemm_flops=100 * 10e9, dt=0.05, traffic=1.8e8, traffic_source="profiles/x.json", gemm_bytes=100 * 8e7)
    assert r["bound"] == "mfma" and abs(r["achieved"] - 200.0) < 1e-9 and abs(r["peak"] - 2500.0 / 3) < 1e-9
    assert abs(r["frac"] - 200.0 / (2500.0 / 3)) < 1e-12 and r["traffic"] == 1.8e8 and r["launches"] == 500 and r["launches_timed"] == 100
    assert abs(r["avg_launch_us"] - 50.0) < 1e-9 and abs(r["gemm_share_of_wall"] - (5e-3 / 100 * 500) / 0.05) < 1e-12
    assert abs(r["hbm_view"]["achieved"] - 100 * 8e7 / 5e-3 / 1e9) < 1e-6
    rb = b.roofline_block("bf16", 5.0, 100, 500, 100 * 10e9, 0.05, gemm_bytes=100 * 8e7)
    assert rb["bound"] == "hbm" and abs(rb["frac"] - rb["achieved"] / 8000.0) < 1e-12
    rf = b.roofline_block("f32", 5.0, 100, 500, 100 * 10e9, 0.05)
    assert rf["peak"] == b.PEAK_F32_MFMA_TFLOPS and rf["bound"] == "mfma"
