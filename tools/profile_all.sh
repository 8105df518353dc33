#!/bin/bash
# Kernel trace and counter passes of every bench.py workload, into gpurun_out/<dir>
# (default r3prof).
set -e
DIR=${1:-r3prof}
for w in deep16m flat1m flat32m probes64; do
  timeout -k 10 300 bash tools/kernel_trace.sh gpurun_out/$DIR/kt_$w --workload $w > /dev/null 2>&1
  echo "trace $w done"
  timeout -k 10 900 bash tools/pmc_passes.sh gpurun_out/$DIR/pmc_$w --workload $w > /dev/null 2>&1
  echo "pmc $w done"
done
