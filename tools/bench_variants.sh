#!/bin/bash
# The secondary bench.py configurations quoted in the design notes, one line each.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
: > "$OUT"
run() { echo "## $*" >> "$OUT"; timeout -k 10 300 python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary --no-dropin --steps 50 --warmup 5 >> "$OUT" 2>&1; }
run --workload flat1m --dirty local
run --workload flat32m --dirty local
run --workload deep16m --dirty local
run --workload flat32m --submeshes 4
run --workload flat32m --instances 134217728
run --workload deep16m --exchange
python3 - "$OUT" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("## "):
        tag = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print(f'{tag:50s} {d["ms_per_step"]*1e3:9.1f} us/frame {d["value"]:.3e}  {r["kernel"]} {r["avg_launch_us"]:8.1f} us frac {r["frac"]}')
PY
