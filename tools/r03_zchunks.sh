#!/bin/bash
# r03: plane-chunked stage tails (KW_FUSED_ZCHUNKS) at grid sizes whose spectral scratch overflows the 256 MB Infinity
# Cache; one box, interleaved repetitions
out=gpurun_out/r03_zchunks.txt
: > $out
for rep in 1 2; do
for n in 512 384 320; do
  for z in 1 2 4 8 16; do
    KW_FUSED_ZCHUNKS=$z timeout -k 10 200 python bench.py --size $n --no-cpu --no-512 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('n=$n zchunks=$z rep=$rep', d['value'], d['ms_per_step'], r['step'].get('frac'), {k.replace('k_',''): round(1e3*v['avg_ms'],1) for k,v in sorted(r['kernels'].items())})
" >> $out || echo "n=$n z=$z FAILED" >> $out
  done
done
done
cat $out
