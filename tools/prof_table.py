#!/usr/bin/env python3
"""kernel table of gpurun_out/r03_prof<n>/ (rocprofv3 --kernel-trace --stats of bench.py --size n): python tools/prof_table.py 64 128"""
import csv, glob, json, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()
for n in sys.argv[1:]:
    f = glob.glob(f"{ROOT}/gpurun_out/r03_prof{n}/**/*kernel_stats.csv", recursive=True)[0]
    line = [l for l in open(f"{ROOT}/gpurun_out/r03_prof{n}.log").read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    steps = d["steps"] + d["warmup"]
    print(f"== {n}^3: {d['value']} steps/s, {d['ms_per_step']} ms/step (profiled run)")
    tot = 0.0
    for r in csv.DictReader(open(f)):
        calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        if calls >= steps * 0.9:
            print(f"  {short(r['Name'])[:56]:56s} x{calls / steps:4.1f}/step  avg {avg:7.2f} us  min {float(r['MinNs']) / 1e3:7.2f}  max {float(r['MaxNs']) / 1e3:8.2f}")
            tot += avg * calls / steps
    print(f"  sum of kernel time per step: {tot:.1f} us")
