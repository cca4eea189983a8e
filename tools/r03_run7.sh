#!/bin/bash
for ex in native p2p; do
  python bench.py --slab-selftest --exchange $ex --steps 30 --warmup 5 --no-512 --no-p2p 2>/dev/null | tail -1 > gpurun_out/r03_slabself_$ex.json
  python tools/slab_host_time.py 256 30 $ex 2>/dev/null | tail -1
done
python -c "
import json
for ex in ('native','p2p'):
    d=json.load(open('gpurun_out/r03_slabself_%s.json'%ex)); print(ex, d['value'], d['ms_per_step'], d['config']['exchanges_per_step'])
"
for m in "75 3" "50 3"; do set -- $m
  for tr in p2p rccl; do
    for e in "" "slab_batch=0,slab_chunks=2"; do
      KW_TUNING="$e" timeout -k 10 200 python tools/emulate_rank.py --transport $tr --grid 512 --ranks 8 --rank 1 --steps 15 --link-gbs $1 --latency-us $2 2>&1 | tail -1 | cut -c1-220
    done
  done
done
for g in "256 8" "256 4" "256 2"; do set -- $g
  for tr in p2p rccl; do
    timeout -k 10 200 python tools/emulate_rank.py --transport $tr --grid $1 --ranks $2 --rank 1 --steps 15 --link-gbs 60 --latency-us 3 2>&1 | tail -1 | cut -c1-220
  done
done
