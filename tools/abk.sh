#!/bin/bash
# tools/abk.sh [bench args]: per-kernel A/B of the device-library builds in ab/<name>/ (build.py --variant) on ONE GPU
# box, two interleaved repetitions; prints steps/s and the kernel table (ms per launch) of every run.  A file ab/<name>/args
# holds extra bench arguments of that build (e.g. "--size 600" for a KW_VARIANT_LENGTH=600 build).
L=k-wave-fluid-cuda_amd/lib
cp $L/libkwave_hip.so /tmp/libkwave_hip.keep
for rep in $(seq 1 ${REPS:-2}); do
  for d in ab/${ABK_GLOB:-*}/; do
    v=$(basename $d)
    cp ab/$v/libkwave_hip.so $L/
    python bench.py --no-cpu --no-512 "$@" $(cat ab/$v/args 2>/dev/null) > gpurun_out/abk_${v}_${rep}.json 2> gpurun_out/abk_${v}_${rep}.err || { echo "$v FAILED"; tail -3 gpurun_out/abk_${v}_${rep}.err; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/abk_${v}_${rep}.json"))
print("$v", "$rep", d["value"], {k.replace("k_",""): round(1e3*x["avg_ms"],1) for k,x in sorted(d["roofline"]["kernels"].items())})
PY
  done
done
cp /tmp/libkwave_hip.keep $L/libkwave_hip.so
