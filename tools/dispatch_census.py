#!/usr/bin/env python
"""Who issues the small dispatches of a step?  (~150 fillBufferAligned + ~85 copyBuffer per step)

    rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1
    python tools/dispatch_census.py DIR/**/trace_kernel_trace.csv [--steps 4]

Reads the kernel trace in start order and, for every runtime fill / copy kernel (and every kernel shorter than 3 us), prints the
name of the next LONG kernel on the same queue: that is the launcher the dispatch belongs to (a memset in front of a kernel that
accumulates into its output).  Prints a table: (kind, next kernel) -> dispatches per step."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
    rows = list(csv.DictReader(open(path)))
    key = lambda r: int(r["Start_Timestamp"])
    rows.sort(key=key)
    by_queue = collections.defaultdict(list)
    for r in rows:
        by_queue[(r.get("Queue_Id"), r.get("Stream_Id"))].append(r)
    small = ("__amd_rocclr_fillBuffer", "__amd_rocclr_copyBuffer")
    census = collections.Counter()
    total = collections.Counter()
    for q, rs in by_queue.items():
        for i, r in enumerate(rs):
            n = r["Kernel_Name"]
            if not n.startswith(small):
                continue
            kind = "fill" if "fill" in n else "copy"
            nxt = "(end of queue)"
            for r2 in rs[i + 1:]:
                if not r2["Kernel_Name"].startswith(small):
                    nxt = short(r2["Kernel_Name"])
                    break
            census[(kind, nxt)] += 1
            total[kind] += 1
    # inside the steps proper: the windows between consecutive starts of the step's first big kernel (--anchor, default cnn1's moments pass)
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "nhwc_first_moments_kernel"
    marks = [key(r) for r in rows if anchor in r["Kernel_Name"]]
    if len(marks) >= 2:
        per = []
        for a, b in zip(marks[:-1], marks[1:]):
            w = [r for r in rows if a <= key(r) < b]
            per.append((len(w), sum("fillBuffer" in r["Kernel_Name"] for r in w), sum("copyBuffer" in r["Kernel_Name"] for r in w)))
        print("inside a step (between two starts of %s): " % anchor
              + "; ".join(f"{n} dispatches, {f} fills, {c} copies" for n, f, c in per))
    print(f"whole process: {len(rows)} dispatches on {len(by_queue)} queues, "
          + ", ".join(f"{v} {k} kernels" for k, v in total.items()) + f"  (the table below: whole process / {steps})")
    for (kind, nxt), c in sorted(census.items(), key=lambda kv: -kv[1]):
        print(f"  {c / steps:7.2f}  {kind}  -> {nxt}")


if __name__ == "__main__":
    main()
