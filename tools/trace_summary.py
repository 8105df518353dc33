#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per-kernel statistics, the
timeline (start offset, duration, gap to the previous kernel) of the last
complete frame, a frame being the kernels from one k_propagate_* / k_classify
launch sequence to the next, and the last kernels of the run with their queues
(kernels of two streams overlap there)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_\w+)", name)
    base = m.group(1) if m else name[:48]
    t = re.search(r"<([^>]*)>", name)
    return base + (f"<{t.group(1)}>" if t else "")


def main() -> int:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                             r.get("Queue_Id", "0")))
    rows.sort()
    queues = {q: i + 1 for i, q in enumerate(sorted({r[3] for r in rows}))}
    timeline = rows[-int(os.environ.get("TIMELINE_KERNELS", "48")):]
    rows = [r[:3] for r in rows]
    stats = defaultdict(list)
    for s, e, k in rows:
        stats[k].append((e - s) / 1e3)
    total = sum(sum(v) for v in stats.values())
    print(f"{'kernel':58s} {'calls':>6s} {'mean_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:58s} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100*sum(v)/total:5.1f}%")
    # last frame: from the last first-level launch of the frame's head kernel
    heads = [i for i, (_, _, k) in enumerate(rows)
             if k.startswith(("k_propagate", "k_classify")) and (i == 0 or not rows[i - 1][2].startswith("k_propagate"))]
    if len(heads) >= 2:
        a, b = heads[-2], heads[-1]
        frame = rows[a:b]
        t0 = frame[0][0]
        print("\nlast complete frame (us from its first kernel):")
        prev_end = None
        for s, e, k in frame:
            gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
            print(f"  {(s - t0)/1e3:9.2f}  dur {(e - s)/1e3:8.2f}  gap {gap:6.2f}  {k}")
            prev_end = e
        print(f"  frame period {(rows[b][0] - t0)/1e3:.2f} us")
    if timeline:
        t0 = timeline[0][0]
        print("\ntimeline of the last kernels (start, end, duration in us, queue, kernel):")
        for s, e, k, q in timeline:
            print(f"  {(s - t0)/1e3:9.1f} {(e - t0)/1e3:9.1f} {(e - s)/1e3:8.1f} q{queues[q]} {k[:60]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
