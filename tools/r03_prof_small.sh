#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in ${SIZES:-64}; do
  rm -rf $R/gpurun_out/r03_prof$n
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_prof$n --output-format csv -- python3 $R/bench.py --size $n --steps 300 --warmup 20 --no-cpu --no-512 > $R/gpurun_out/r03_prof$n.log 2>&1 || exit 1
done
cd $R && python3 tools/prof_table.py ${SIZES:-64}
