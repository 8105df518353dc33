#!/bin/bash
# Where the deep16m frame differs from a flat one: hierarchy and view count varied separately.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
: > "$OUT"
run() { echo "## $*" >> "$OUT"; timeout -k 10 200 python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary --no-dropin --steps 50 --warmup 5 >> "$OUT" 2>&1; }
run --workload deep16m
run --workload deep16m --cascades 0
run --workload flat32m --instances 16777216
run --workload flat32m --instances 16777216 --cascades 4
python3 - "$OUT" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("## "):
        tag = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print(f'{tag:60s} {d["ms_per_step"]*1e3:9.1f} us/frame  fused {r["avg_launch_us"]:8.1f} us  survivors/bytes {r["algorithmic_bytes_per_step"]}')
PY
