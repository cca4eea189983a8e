#!/usr/bin/env python3
"""Per-kernel table of bench.py result lines:  python tools/bench_table.py line1.json [line2.json ...]
(name, us per launch, launches per step, GB/s of the kernel's own algorithmic bytes)"""
import json
import sys

for path in sys.argv[1:]:
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    if not lines:
        print(path, ": no result line")
        continue
    d = json.loads(lines[-1])
    r = d["roofline"]
    print(f"== {path}: {d['value']} {d['unit']}, {d['ms_per_step']} ms/step, step {r.get('step')}")
    for k, v in sorted(r.get("kernels", {}).items()):
        print(f"  {k:28s} {v['avg_ms'] * 1e3:9.1f} us x {v['calls_per_step']:.1f}  {v.get('alg_gbs', 0):8.1f} GB/s")
