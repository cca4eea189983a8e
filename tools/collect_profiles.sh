#!/bin/bash
# Runs ON the GPU box (gpurun -- tools/collect_profiles.sh <tag>): rocprofv3 kernel statistics and PMC traffic counters
# of the bench workloads, each in its own run, into gpurun_out/<tag>/.  tools/summarize_profiles.py condenses them
# into profiles/.
tag=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -x
(cd $R && python3 -c "import bench; print(bench.kernel_source_hash())") > $O/source_hash.txt
rocprofv3 --kernel-trace --stats -d $O/stats256 --output-format csv -- python3 $R/bench.py --steps 40 --warmup 3 --no-cpu --no-512 > $O/stats256.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc256_$c --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-512 --profile-steps 1 > $O/pmc256_$c.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc $c -d $O/pmcprobe_$c --output-format csv -- python3 $R/tools/probe_passes.py 256 3 > $O/pmcprobe_$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc256_SQ --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-512 --profile-steps 1 > $O/pmc256_SQ.log 2>&1 || echo "SQ counters not collected"
rocprofv3 --kernel-trace --stats -d $O/stats512 --output-format csv -- python3 $R/bench.py --size 512 --steps 10 --warmup 2 --no-cpu --no-512 > $O/stats512.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/statsslab --output-format csv -- python3 $R/bench.py --slab-selftest --steps 20 --warmup 3 --no-p2p > $O/statsslab.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/statsslab_p2p --output-format csv -- python3 $R/bench.py --slab-selftest --exchange p2p --steps 20 --warmup 3 > $O/statsslab_p2p.log 2>&1 || exit 1
for n in 128 64; do
  rocprofv3 --kernel-trace --stats -d $O/stats$n --output-format csv -- python3 $R/bench.py --size $n --steps 300 --warmup 20 --no-cpu --no-512 > $O/stats$n.log 2>&1 || exit 1
done
echo collected
