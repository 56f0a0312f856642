#!/bin/bash
# tools/gpu_steps.sh -- run a list of GPU steps on the gpurun box, each under its own `timeout -k 10`, logging to
# gpurun_out/<tag>_<name>.log.  A step that fails (tests red) does not stop the list; a step that TIMES OUT or is killed
# does (no further GPU work after a hang).  Usage: tools/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  log=gpurun_out/${tag}_${name}.log
  echo "== step $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$log" 2>&1
  rc=$?
  echo "== step $name rc=$rc after $(( $(date +%s) - start ))s" | tee -a "$log"
  tail -n 6 "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== step $name timed out / was killed: stopping"; exit $rc; fi
done
exit 0
