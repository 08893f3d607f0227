#!/usr/bin/env python
"""Prints a time window of a rocprofv3 kernel trace, all queues merged: python tools/trace_window.py TRACE.csv ANCHOR_SUBSTRING [before_us after_us [occurrence]]"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    anchor = sys.argv[2]
    before = float(sys.argv[3]) if len(sys.argv) > 3 else 500.0
    after = float(sys.argv[4]) if len(sys.argv) > 4 else 2000.0
    occ = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    hits = [int(r["Start_Timestamp"]) for r in rows if anchor in r["Kernel_Name"]]
    t = hits[occ]
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t - before * 1e3 <= s <= t + after * 1e3:
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n).split("(")[0][:64]
            print(f"{(s - t) / 1e3:9.1f}us q{r['Queue_Id']} {(e - s) / 1e3:8.1f}us  {n}  grid {r['Grid_Size_X']}")


if __name__ == "__main__":
    main()
