#!/bin/bash
# One build under several settings of an environment variable, in turn, on ONE
# configuration given as bench.py arguments.
#   tools/ab_env_multi.sh <out-file> <VAR> "<v1> <v2> ..." <bench.py arguments...>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; VAR=$2; VALUES=$3; shift 3
: > "$OUT"
for rep in $(seq ${REPS:-2}); do
  for value in $VALUES; do
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
