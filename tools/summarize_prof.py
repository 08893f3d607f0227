#!/usr/bin/env python
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh) into the small CSVs kept
under profiles/<tag>_rocprof/: the rocprofv3 --stats kernel table as is, and one row per kernel
with the mean of every PMC counter per dispatch plus the derived figures DESIGN.md quotes.

    python tools/summarize_prof.py r01_train
"""
import collections
import csv
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", f"{tag}_rocprof")
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.isfile(stats):
        shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
    else:      # (rocprofv3 died before its --stats table: the same table from the kernel trace)
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))):
            by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = sum(sum(v) for v in by.values()) or 1
        with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                m = sum(v) / len(v)
                sd = (sum((x - m) ** 2 for x in v) / (len(v) - 1)) ** 0.5 if len(v) > 1 else 0.0
                w.writerow([k, len(v), sum(v), round(m, 6), round(100.0 * sum(v) / total, 4), min(v), max(v), round(sd, 6)])
    if os.path.isfile(os.path.join(src, "sources.sha256")):
        shutil.copy(os.path.join(src, "sources.sha256"), os.path.join(dst, "sources.sha256"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds"):
        path = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.isfile(path):
            continue
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if sub == "pmc_sq" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    counters = sorted({c for v in per.values() for c in v})
    with open(os.path.join(dst, "pmc_per_kernel.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "mean_ms_under_pmc"] + counters +
                   ["mfma_busy_frac", "lds_conflict_frac", "fetch_GB_x2", "write_GB"])
        for k, v in sorted(per.items()):
            mean = {c: (sum(v[c]) / len(v[c]) if v.get(c) else None) for c in counters}
            n = max(len(x) for x in v.values())
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs (1024); GRBM_GUI_ACTIVE counts per-XCD
            # cycles summed over the 8 XCDs  ->  busy fraction = busy / (gui/8 * 1024)
            mf = None
            if mean.get("SQ_VALU_MFMA_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE"):
                mf = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 1024)
            lc = None
            if mean.get("SQ_LDS_IDX_ACTIVE"):
                lc = (mean.get("SQ_LDS_BANK_CONFLICT") or 0.0) / mean["SQ_LDS_IDX_ACTIVE"]
            # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE reports half of a wide streaming
            # read (MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE uncalibrated
            fe = mean["FETCH_SIZE"] * 1024 * 2 / 1e9 if mean.get("FETCH_SIZE") else None
            wr = mean["WRITE_SIZE"] * 1024 / 1e9 if mean.get("WRITE_SIZE") else None
            md = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
            fmt = lambda x: "" if x is None else f"{x:.6g}"
            w.writerow([k, n, fmt(md)] + [fmt(mean[c]) for c in counters] + [fmt(mf), fmt(lc), fmt(fe), fmt(wr)])
    print("wrote", dst, os.listdir(dst))


if __name__ == "__main__":
    main()
