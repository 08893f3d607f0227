"""The committed measurement records of the latest round are what the documents and bench.py's `traffic` lookup say they are
(this test had been deleted instead of moved on; bench.py filters the PMC summary by substrings of the kernels'
template arguments, so a renamed instance would silently turn `traffic` into null)."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
import glob

# the newest round that has a committed default line (bench.py's own counter lookup discovers the newest round the same way)
ROUND = max(os.path.basename(p)[:3] for p in glob.glob(os.path.join(PROF, "r[0-9][0-9]_bench_train.json")))


def _line(name):
    with open(os.path.join(PROF, f"{ROUND}_{name}.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_headline_line_keeps_the_bench_contract():
    d = _line("bench_train")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["batch_per_gpu"] == 64
    # value is utterances / time of exactly `steps` steps
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert len(r["launch_ms"]) == r["launches_per_step"] == 10
    # the dominant kernel's launches fit inside the step
    assert sum(r["launch_ms"]) < d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    if ROUND >= "r05":
        assert "forward_b1" in c                                   # BASELINE configs[0]
        for leg in ("forward", "forward_bf16"):
            assert "roofline" in d[leg] and "stage_ms" in d[leg], leg
        assert all("roofline" in v for v in d["longform"].values())
        assert "tflops" in d["box_calibration"]


@pytest.mark.parametrize("tag,needles", [
    ("train_bf16", [("nhwc_conv_kernel<5, 5", "true, false>"), ("nhwc_conv_kernel<5, 5", "false, true>"), ("nhwc_wgrad_kernel<5, 5", ""),
                    ("gemm_bf16_il_kernel", "")]),
    ("forward", [("nhwc_conv_f16x3", "5, 5")]),
    # round 6, call 24: the legs whose `traffic` used to be null
    ("forward_bf16", [("nhwc_conv_kernel<5, 5", "")]),
    ("longform", [("nhwc_conv_f16x3", "5, 5")]),
    ("longform_bf16", [("nhwc_conv_kernel<5, 5", "")]),
    ("train_f16x3", [("conv64_f16x3_pk_kernel", ""), ("conv64_wgrad_ring4_kernel", "")]),
])
def test_rocprof_summaries_hold_the_rows_bench_py_looks_up(tag, needles):
    d = os.path.join(PROF, f"{ROUND}_{tag}_rocprof")
    if ROUND < "r06" and tag not in ("train_bf16", "forward"):
        pytest.skip("collected from round 6 on")
    stats = os.path.join(d, "kernel_stats.csv")
    pmc = os.path.join(d, "pmc_per_kernel.csv")
    assert os.path.isfile(stats) and os.path.isfile(pmc), d
    names = [r["Name"] for r in csv.DictReader(open(stats))]
    rows = list(csv.DictReader(open(pmc)))
    for kern, inst in needles:
        assert any(kern in n and inst in n for n in names), (kern, inst, "kernel_stats.csv")
        hit = [r for r in rows if kern in r["kernel"] and inst in r["kernel"]]
        assert hit, (kern, inst, "pmc_per_kernel.csv")
        if "conv" in kern:
            assert any(r.get("fetch_GB_x2") and r.get("write_GB") for r in hit), (kern, inst)
    if ROUND >= "r05":
        rec = os.path.join(d, "sources.sha256")
        assert os.path.isfile(rec)
        listed = {ln.split()[1].lstrip("*") for ln in open(rec) if ln.strip()}
        assert {"conv_nhwc.hip", "conv_nhwc_f16x3.hip", "gemm_bf16.hip"} <= listed


@pytest.mark.parametrize("tag,kernel,instance,source", [
    ("train_bf16", "nhwc_conv_kernel<5, 5", "true, false>", "conv_nhwc.hip"),
    ("train_bf16", "nhwc_conv_kernel<5, 5", "false, true>", "conv_nhwc.hip"),
    ("forward", "nhwc_conv_f16x3", "<5, 5", "conv_nhwc_f16x3.hip"),
    ("forward_bf16", "nhwc_conv_kernel", None, "conv_nhwc.hip"),
    ("longform", "nhwc_conv_f16x3", "<5, 5", "conv_nhwc_f16x3.hip"),
    ("longform_bf16", "nhwc_conv_kernel", None, "conv_nhwc.hip"),
    ("train_f16x3", "conv64_f16x3_pk_kernel", None, "conv_f16x3_pk.hip"),
])
def test_committed_counters_belong_to_the_kernel_sources_of_this_tree(tag, kernel, instance, source):
    """bench.py reports `roofline.traffic` only while the kernel's source file is byte-identical to the one the counters were collected
    on (sources.sha256 beside the summary): a kernel edit without a new profile must show up HERE, not as a silent null in the driver's
    line.  Traffic within 1.0-1.35x of the algorithmic bytes is also what DESIGN.md section 7 states for every leg."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    if ROUND < "r06":
        pytest.skip("every leg has counters from round 6 on")
    gb, why = bench.committed_pmc_traffic(tag, kernel, instance, source=source)
    assert gb is not None, why
    assert why == ROUND, why
    one = 64 * 64 * 301 * 601 / 1e9                      # elements of one [64, 64, 301, 601] activation tensor, in G
    algo = {"train_bf16": (2 if instance and instance.startswith("true") else 3) * one * 2, "forward": 2 * one * 4, "forward_bf16": 2 * one * 2,
            "longform": 4 * 2 * one * 4, "longform_bf16": 4 * 2 * one * 2, "train_f16x3": 2 * one * 4}[tag]
    assert 1.0 <= gb / algo <= 1.35, (gb, algo)
