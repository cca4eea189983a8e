#!/bin/bash
# tools/gpu_steps.sh "<label>|<seconds>|<command>" ...   — run GPU steps one after the other on a gpurun box, each under its own
# timeout, output to gpurun_out/<label>.log.  A step that fails goes on to the next; a step that had to be KILLED (timeout)
# ends the call: nothing else is started on a GPU that may be hung.
mkdir -p gpurun_out
for spec in "$@"; do
  IFS='|' read -r label secs cmd <<< "$spec"
  echo "=== $label (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$label.log" 2>&1
  rc=$?
  echo "EXIT $rc" >> "gpurun_out/$label.log"
  tail -n ${TAIL:-6} "gpurun_out/$label.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $label was killed at its limit: stopping here"; exit 1; fi
done
exit 0
