#!/bin/bash
# r03: kernel traces of the small grids, slab self-test over both transports, link-model schedules
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 64 128; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof$n --output-format csv -- python3 $R/bench.py --size $n --steps 300 --warmup 20 --no-cpu --no-512 > $R/gpurun_out/r03_prof$n.log 2>&1 || exit 1
done
cd $R
for ex in native p2p; do
  python bench.py --slab-selftest --exchange $ex --steps 30 --warmup 5 --no-512 --no-p2p 2>/dev/null | tail -1 > gpurun_out/r03_slabself_$ex.json
  python tools/slab_host_time.py 256 30 $ex 2>/dev/null | tail -1
done
python -c "
import json
for ex in ('native','p2p'):
    d=json.load(open('gpurun_out/r03_slabself_%s.json'%ex)); print(ex, d['value'], d['ms_per_step'], d['config']['exchanges_per_step'])
"
CASES="512 8" MODELS="60 3;60 10;60 30" TRANSPORTS="p2p rccl" bash tools/emulate_schedules.sh
