#!/usr/bin/env python3
"""Sums rocprofv3 counter-collection CSVs per kernel and counter.

    tools/pmc_summary.py <dir with one sub-directory per pass> [<frames> <traffic.json>]

Prints, per kernel, the number of dispatches seen and the mean value per
dispatch of every counter. FETCH_SIZE / WRITE_SIZE are reported in bytes
(rocprofv3 gives KiB); FETCH_SIZE is also shown doubled, the correction gfx950
needs for wide coalesced reads (128-byte requests are tallied as 64 bytes).
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_\w+)", name)
    base = m.group(1) if m else name[:40]
    t = re.search(r"<([^>]*)>", name)
    return base + (f"<{t.group(1)}>" if t else "")


def main() -> int:
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[k][c][0] += float(row["Counter_Value"])
                acc[k][c][1] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            total, n = acc[k][c]
            mean = total / max(n, 1)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                b = mean * 1024.0
                extra = f"  (x2 = {2 * b:.4g} B)" if c == "FETCH_SIZE" else ""
                print(f"  {c:24s} n={n:5d} mean={b:.6g} B{extra}")
            else:
                print(f"  {c:24s} n={n:5d} mean={mean:.6g}")
    if len(sys.argv) > 3:  # <frames> <json path>: HBM traffic per kernel for bench.py
        import json
        frames = int(sys.argv[2])
        out = {"frames": frames, "kernels": {}}
        for k in acc:
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                f_total, f_n = acc[k]["FETCH_SIZE"]
                w_total, w_n = acc[k]["WRITE_SIZE"]
                out["kernels"][k] = {
                    "dispatches_per_frame": f_n / frames,
                    "fetch_bytes_per_dispatch_x2": 2 * 1024.0 * f_total / max(f_n, 1),
                    "write_bytes_per_dispatch": 1024.0 * w_total / max(w_n, 1),
                }
                for counter, key in (("SQ_INSTS_VALU", "valu_insts_per_dispatch"),
                                     ("SQ_INSTS_SALU", "salu_insts_per_dispatch")):
                    if counter in acc[k]:  # wave instructions, all shader engines summed
                        total, n = acc[k][counter]
                        out["kernels"][k][key] = total / max(n, 1)
        with open(sys.argv[3], "w") as fh:
            json.dump(out, fh, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
