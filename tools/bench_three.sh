#!/bin/bash
# bench.py on the four workloads, one JSON line each (no CPU baseline).
#   tools/bench_three.sh <out-file>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
: > "$OUT"
for w in flat1m deep16m flat32m probes64; do
  timeout -k 10 200 python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
done
python3 - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(f'{d["config"]["workload"][:9]:10s} {d["ms_per_step"]*1e3:9.1f} us/frame  {d["value"]:.3e} inst/s  '
              f'{r["kernel"]} {r["avg_launch_us"]:8.1f} us  frac {r["frac"]}')
PY
