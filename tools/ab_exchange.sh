#!/bin/bash
# A/B of two builds on the frame with the exchange step (one rank).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
: > "$OUT"
for rep in 1 2; do
  for which in ${VARIANTS:-base new}; do
    cp "$ROOT/ab/libvkrhip_$which.so" "$ROOT/vulkan_renderer_amd/libvkrhip.so"
    for n in 1048576 4194304; do
      echo "## $which n=$n" >> "$OUT"
      timeout -k 10 200 python3 "$ROOT/bench.py" --instances $n --exchange --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
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
        d = json.loads(line)
        print(f'{tag:30s} {d["ms_per_step"]*1e3:9.1f} us/frame')
PY
