#!/usr/bin/env python
"""Print the mean PMC counters of one kernel from gpurun_out/pmc16_{a,b,c} (ad-hoc profiling of
tools/conv_micro.py) with the derived ratios used while tuning."""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key = sys.argv[1] if len(sys.argv) > 1 else "ILi5ELi5ELi2ELi1E"
tot = {}
for sub in "abc":
    path = os.path.join(ROOT, "gpurun_out", f"pmc16_{sub}", "pmc_counter_collection.csv")
    if not os.path.isfile(path):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            tot["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for c, x in acc.items():
        tot[c] = sum(x) / len(x)
for k, v in tot.items():
    print(f"{k:32s} {v:.4g}")
g = tot["GRBM_GUI_ACTIVE"] / 8
print("clock GHz", g / tot["dur_ms"] / 1e6)
print("MFMA busy frac", tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024))
print("WAIT_INST_ANY/WAVE", tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], "WAIT_ANY/WAVE", tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
      "ACTIVE/WAVE", tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"])
if "SQ_LDS_IDX_ACTIVE" in tot:
    print("LDS active frac/CU", tot["SQ_LDS_IDX_ACTIVE"] / (g * 256), "conflict frac", tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"])
    print("VALU insts per wave", tot["SQ_INSTS_VALU"] / tot.get("SQ_WAVES", 1))
