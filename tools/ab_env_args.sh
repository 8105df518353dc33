#!/bin/bash
# A/B of one build under two settings of an environment variable, alternating, on
# ONE configuration given as bench.py arguments.
#   tools/ab_env_args.sh <out-file> <VAR> <value-a> <value-b> <bench.py arguments...>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; VAR=$2; A=$3; B=$4; shift 4
: > "$OUT"
for rep in $(seq ${REPS:-2}); do
  for value in $A $B; do
    echo "## $VAR=$value $*" >> "$OUT"
    env $VAR=$value timeout -k 10 200 python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
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
        print(f'{tag:60s} {d["ms_per_step"]*1e3:9.1f} us/frame  fused {r["avg_launch_us"]:8.1f} us')
PY
