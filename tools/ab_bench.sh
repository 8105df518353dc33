#!/bin/bash
# A/B of two builds of the library on one box: ab/libvkrhip_base.so against
# ab/libvkrhip_new.so, alternating, on the given workloads.
#   tools/ab_bench.sh <out-file> <workload>...
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; shift
: > "$OUT"
for rep in $(seq ${REPS:-3}); do
  for which in ${VARIANTS:-base new}; do
    cp "$ROOT/ab/libvkrhip_$which.so" "$ROOT/vulkan_renderer_amd/libvkrhip.so"
    for w in "$@"; do
      echo "## $which $w" >> "$OUT"
      timeout -k 10 200 python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline --no-secondary --no-dropin >> "$OUT" 2>&1
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
        print(f'{tag:16s} {d["ms_per_step"]*1e3:9.1f} us/frame  fused {r["avg_launch_us"]:8.1f} us')
PY
