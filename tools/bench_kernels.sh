#!/bin/bash
# per-kernel table of bench.py at the given cube sizes: name, ms per launch, algorithmic GB/s
for n in ${SIZES:-500 600}; do
  echo "== $n"
  timeout -k 10 300 python bench.py --size $n --no-cpu --steps ${STEPS:-30} --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['step'])
for k,v in sorted(r['kernels'].items()): print(f'  {k:28s} {v[\"avg_ms\"]*1e3:9.1f} us x {v[\"calls_per_step\"]:.1f}  {v[\"alg_gbs\"]:8.1f} GB/s')
" || exit 1
done
