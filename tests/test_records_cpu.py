"""The committed measurement records of the round (profiles/) are well-formed: the bench lines carry the contract fields
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / dtype / data / config), the
`roofline` object is consistent with itself (frac = achieved / peak, the per-launch times it was derived from are there)
and `cpu_baseline` says what was timed; the rocprofv3 summaries named by profiles/README.md exist."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
ROUND = "r03"


def _line(name):
    path = os.path.join(PROF, f"{ROUND}_{name}.json")
    assert os.path.isfile(path), path
    return json.loads(open(path).read().strip().splitlines()[-1])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


@pytest.mark.parametrize("name", ["bench_train", "bench_forward", "bench_forward_bf16", "bench_longform", "bench_longform_bf16",
                                  "bench_train_serial_backward", "bench_train_voicefilter_powerlaw", "bench_train_b2"])
def test_bench_lines_carry_the_contract(name):
    d = _line(name)
    for k in CONTRACT:
        assert k in d, (name, k)
    assert d["unit"] == "utterances/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                       # BASELINE.md holds no published number for this metric
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    # value is the whole-job rate of the timed region: utterances (windows) per step / time per step
    per_step = d["config"].get("global_batch")
    if "longform" in name:                                # a step = 128 clips of 30 s = 1280 windows (256 per forward batch)
        per_step = 128 * 10
    if per_step:
        assert abs(d["value"] - per_step / d["ms_per_step"] * 1e3) <= 0.02 * d["value"], (name, d["value"], per_step, d["ms_per_step"])


def test_headline_roofline_and_cpu_baseline():
    d = _line("bench_train")
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    ms = r["launch_ms"]
    assert len(ms) == r["launches_per_step"] == 10 and all(m > 0 for m in ms)
    # achieved = algorithmic FLOPs of one launch (64 utterances x 37.05 GFLOP) / mean launch time
    assert abs(r["achieved"] - 64 * 37.05 / (sum(ms) / len(ms))) < 0.02 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 2.9      # GB per launch; algorithmic 2.96
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "utterances/s"
    for sub in ("forward", "forward_bf16", "fp32_class", "fp32_strict", "rccl"):
        assert sub in d, sub
    assert d["dtype"].startswith("bf16")


def test_rocprof_summaries_are_there():
    base = os.path.join(PROF, f"{ROUND}_train_bf16_rocprof")
    rows = list(csv.DictReader(open(os.path.join(base, "kernel_stats.csv"))))
    names = " ".join(r["Name"] for r in rows)
    for k in ("nhwc_conv_kernel<5, 5", "nhwc_wgrad_kernel<5, 5", "lstm16_persistent_kernel", "lstm16_bwd_persistent_kernel", "gemm_bf16_kernel"):
        assert k in names, k
    pmc = list(csv.DictReader(open(os.path.join(base, "pmc_per_kernel.csv"))))
    conv = [r for r in pmc if "nhwc_conv_kernel<5, 5" in r["kernel"]]
    assert conv and all(float(r["fetch_GB_x2"]) > 1.4 and float(r["write_GB"]) > 1.4 for r in conv)
    for f in (f"{ROUND}_pytest_gpu.log", f"{ROUND}_step_timeline.txt", f"{ROUND}_lstm_time.txt", f"{ROUND}_probes.txt"):
        assert os.path.getsize(os.path.join(PROF, f)) > 0, f
    assert "passed" in open(os.path.join(PROF, f"{ROUND}_pytest_gpu.log")).read()
