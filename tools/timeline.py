"""Prints the kernel timeline (start, end, duration in us, queue, kernel) of the last
frames in a rocprofv3 --kernel-trace CSV: python tools/timeline.py <csv> [frames]."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame starts with the merged top-level launch (the fifth template flag of the fused kernel)
top = re.compile(r"k_propagate_classify<\w+, \w+, \w+, \w+, true")
starts = [i for i, r in enumerate(rows) if top.search(r["Kernel_Name"])]
first = starts[-(frames + 1)] if len(starts) > frames else 0
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    n = r["Kernel_Name"]
    n = n.replace("vkr::", "").replace("void ", "")[:58]
    b, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{b / 1e3:9.1f} {e / 1e3:9.1f} {(e - b) / 1e3:8.1f} q{r['Queue_Id']} {n}")
