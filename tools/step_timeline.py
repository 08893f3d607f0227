#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every dispatch with its queue, start offset and
duration, the union of busy time, and the idle gaps of the device (no kernel running on any queue).

    rocprofv3 --kernel-trace -d out -o trace -f csv -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline
    python tools/step_timeline.py out/.../trace_kernel_trace.csv [marker substring, default nhwc_conv_first] > timeline.txt
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:60]


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "nhwc_conv_first"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[3]]
    if len(marks) < 2:
        print("marker not found twice:", marker, len(marks))
        return
    lo, hi = marks[-2], marks[-1]            # the last complete step
    step = rows[lo:hi]
    t0 = step[0][0]
    span = rows[hi][0] - t0
    queues = {q: i for i, q in enumerate(sorted({r[2] for r in step}))}
    busy, gaps, cur_end = 0, [], t0
    for s, e, q, n in step:
        if s > cur_end:
            gaps.append((cur_end - t0, s - cur_end, n))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    if rows[hi][0] > cur_end:
        gaps.append((cur_end - t0, rows[hi][0] - cur_end, "<next step>"))
    print(f"step span {span / 1e6:.3f} ms, {len(step)} dispatches on {len(queues)} queues, device busy (union) {busy / 1e6:.3f} ms, "
          f"idle {sum(g[1] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
    print("largest gaps (offset ms, gap us, next kernel):")
    for off, g, n in sorted(gaps, key=lambda x: -x[1])[:25]:
        print(f"  {off / 1e6:9.3f} {g / 1e3:9.1f}  {n}")
    hist = [0, 0, 0, 0]
    for _, g, _ in gaps:
        hist[0 if g < 5e3 else 1 if g < 20e3 else 2 if g < 100e3 else 3] += g
    print("idle by gap size: <5us %.3f ms, 5-20us %.3f ms, 20-100us %.3f ms, >100us %.3f ms" % tuple(h / 1e6 for h in hist))
    print("timeline (offset ms, dur us, queue, kernel):")
    for s, e, q, n in step:
        print(f"  {(s - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f}  q{queues[q]}  {n}")


if __name__ == "__main__":
    main()
