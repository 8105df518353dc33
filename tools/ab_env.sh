#!/bin/bash
# A/B of one build under two settings of an environment variable, alternating.
#   tools/ab_env.sh <out-file> <VAR> <value-a> <value-b> <workload>...
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; VAR=$2; A=$3; B=$4; shift 4
: > "$OUT"
for rep in $(seq ${REPS:-2}); do
  for value in $A $B; do
    for w in "$@"; do
      echo "## $VAR=$value $w" >> "$OUT"
      env $VAR=$value timeout -k 10 200 python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
    done
  done
done
python3 - "$OUT" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("## "):
        tag = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line); r = d["roofline"]
        print(f'{tag:34s} {d["ms_per_step"]*1e3:9.1f} us/frame  fused {r["avg_launch_us"]:8.1f} us')
PY
