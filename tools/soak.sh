#!/bin/bash
# usage: tools/soak.sh N : the multi-process GPU tests N times in a row on one box
mkdir -p gpurun_out/r06
: > gpurun_out/r06/soak_final.log
for i in $(seq 1 $1); do
  echo "=== run $i $(date +%T)" >> gpurun_out/r06/soak_final.log
  timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_gpu_persistent_guard.py -q -m gpu -k "two_ranks or sharded or independent or eight_ranks or forced" 2>&1 | grep -v "^$" | tail -12 >> gpurun_out/r06/soak_final.log
done
[ -f gpurun_out/shared_gpu_restarts.log ] && { echo "=== restarts"; cat gpurun_out/shared_gpu_restarts.log; } >> gpurun_out/r06/soak_final.log
grep -c "passed" gpurun_out/r06/soak_final.log; grep "passed\|failed\|restart" gpurun_out/r06/soak_final.log | head -20
