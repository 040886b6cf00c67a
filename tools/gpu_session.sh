#!/bin/bash
# Run a list of GPU steps on the gpurun box; each step under its own timeout; stop at the first step that
# was killed by its timeout (a hung kernel), but carry on after ordinary failures (e.g. failing asserts).
# usage: tools/gpu_session.sh "name|timeout_s|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name TIMED OUT: stopping the session"; exit 1; fi
done
exit 0
