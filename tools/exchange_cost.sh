#!/bin/bash
# What the exchange step adds to a frame at one rank, for three frame lengths.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
: > "$OUT"
for n in 1048576 4194304 16777216; do
  for x in "" "--exchange"; do
    echo "## n=$n $x" >> "$OUT"
    timeout -k 10 200 python3 "$ROOT/bench.py" --instances $n $x --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("## "):
        tag = line[3:].strip()
    elif line.startswith("{"):
        d = json.loads(line)
        print(f'{tag:30s} {d["ms_per_step"]*1e3:9.1f} us/frame')
PY
