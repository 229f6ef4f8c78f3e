#!/bin/bash
# Run GPU steps one after another on the gpurun box; stop the whole session as soon as one step
# times out or is killed (never start another GPU step after a hang).
# usage: tools/gpu_steps.sh "<secs>|<outfile>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  secs="${spec%%|*}"; rest="${spec#*|}"; out="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$(date +%T)] $cmd (limit ${secs}s) -> gpurun_out/$out"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$out" 2>&1
  rc=$?
  echo "    rc=$rc"
  tail -n 3 "gpurun_out/$out" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out/killed: stopping session"; exit 1; fi
done
exit 0
