#!/bin/bash
# Runs tools/gloo_probe.py in every mode, each under its own time limit; a mode
# that stalls ends with its stack dump and the next mode still runs (nothing here
# can hang the GPU: a stalled rank waits on the host and is killed by its timer).
#   tools/gloo_probe.sh <out-dir>
OUT=$1
mkdir -p "$OUT"
port=29540
for mode in ${MODES:-plain busy side staged allreduce gatherlist}; do
  port=$((port + 1))
  timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $port tools/gloo_probe.py $mode 300 > "$OUT/$mode.log" 2>&1
  echo "$mode: exit $?" | tee -a "$OUT/summary.txt"
  grep -E "^\[$mode\]|Timeout|all_gather|synchronize|File \"/usr" "$OUT/$mode.log" | tail -12 >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
