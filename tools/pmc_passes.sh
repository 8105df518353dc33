#!/bin/bash
# Hardware-counter passes over one bench.py configuration, one rocprofv3 run per
# counter group (the SQ and TCC blocks cannot hold more in a single run). Each pass
# runs 1 warm-up + 3 timed frames = 4 frames of the full path (no settling frames:
# the summary divides the dispatch counts by that number).
#   tools/pmc_passes.sh <out-dir> <bench.py arguments...>
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
export VKR_BENCH_SETTLE_S=0
run_pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$ROOT/$OUT/$name" --output-format csv -- \
    python3 "$ROOT/bench.py" $BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-dropin \
    > "$ROOT/$OUT/$name.log" 2>&1
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
run_pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" 4 "$ROOT/$OUT/traffic.json" > "$ROOT/$OUT/summary.txt"
cat "$ROOT/$OUT/summary.txt"
