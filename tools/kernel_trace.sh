#!/bin/bash
# Kernel trace of one bench.py configuration with rocprofv3, summarised per
# kernel (count, mean, min) and as a timeline of one steady-state frame.
#   tools/kernel_trace.sh <out-dir> <bench.py arguments...>
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace" --output-format csv -- \
  python3 "$ROOT/bench.py" "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-dropin > "$ROOT/$OUT/bench.log" 2>&1
python3 "$ROOT/tools/trace_summary.py" "$ROOT/$OUT/trace" > "$ROOT/$OUT/summary.txt"
cat "$ROOT/$OUT/summary.txt"
