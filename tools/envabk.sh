#!/bin/bash
# tools/envabk.sh VAR "v1 v2 ..." [bench args]: per-kernel A/B of one build under VAR=v for each value on ONE GPU box,
# two interleaved repetitions; prints steps/s and the kernel table (us per launch) of every run.
var=$1; vals=$2; shift; shift
for rep in 1 2; do
  for v in $vals; do
    env $var=$v python bench.py --no-cpu --no-512 "$@" > gpurun_out/envk_${var}_${v}_${rep}.json 2> gpurun_out/envk_${var}_${v}_${rep}.err || { echo "$var=$v FAILED"; tail -3 gpurun_out/envk_${var}_${v}_${rep}.err; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/envk_${var}_${v}_${rep}.json"))
print("$var=$v", "$rep", d["value"], {k.replace("k_",""): round(1e3*x["avg_ms"],1) for k,x in sorted(d["roofline"]["kernels"].items())})
PY
  done
done
