#!/bin/bash
# Runs ON the GPU box: (1) per-kernel A/B of the builds in ab/ (tools/abk.sh), (2) the one-rank slab path under rocprofv3 with the
# serial schedule (slab_pipeline=0: multi-array launches, nothing overlapping the exchanges) beside the default one
R=$GRAFT_REPO_ROOT
cd $R && bash tools/abk.sh > gpurun_out/r03_abk_blocks.txt 2>&1
cat gpurun_out/r03_abk_blocks.txt
cd /tmp && export TMPDIR=/tmp
for sched in 0 1; do
  rm -rf $R/gpurun_out/r03_slab_sched$sched
  KW_TUNING=slab_pipeline=$sched rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_slab_sched$sched --output-format csv -- python3 $R/bench.py --slab-selftest --exchange p2p --steps 20 --warmup 3 > $R/gpurun_out/r03_slab_sched$sched.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/r03_slab_sched$sched.log | cut -c1-400
done
