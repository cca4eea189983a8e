#!/bin/bash
# tools/envab.sh VAR "v1 v2 ..." [bench args] : bench the current build under VAR=v for each value, two interleaved
# repetitions on one box
var=$1; vals=$2; shift; shift
for rep in 1 2; do
  for v in $vals; do
    env $var=$v python bench.py --no-cpu "$@" > gpurun_out/env_${var}_${v}_${rep}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/env_${var}_${v}_${rep}.json"))
print("$var=$v", "$rep", d["value"], {k: round(x["ms_per_step"],4) for k,x in d["roofline"]["entry_points"].items()})
PY
  done
done
