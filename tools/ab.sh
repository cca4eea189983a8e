#!/bin/bash
# Compare builds of the native libraries on ONE GPU box (box-to-box spread is ~6 %, larger than most tuning steps):
#   tools/ab.sh [bench args]      libraries in ab/<name>/*.so, two interleaved repetitions each
set -e
L=k-wave-fluid-cuda_amd/lib
for rep in 1 2; do
  for d in ab/*/; do
    v=$(basename $d)
    cp ab/$v/libkwave_hip.so $L/   # device layer only: the host layer of the working tree binds newer kwh_* symbols
    python bench.py --no-cpu "$@" > gpurun_out/ab_${v}_${rep}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_${v}_${rep}.json"))
print("$v", "$rep", d["value"], {k: round(x["ms_per_step"],4) for k,x in d["roofline"]["entry_points"].items()})
PY
  done
done
