#!/bin/bash
# Runs ON the GPU box: SQ / LDS counters of the fused kernels at several cube sizes (one rocprofv3 --pmc pass per counter set),
# then tools/pmc_len_table.py prints them per kernel.  SIZES="240 256" bash tools/r03_pmc_len.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in ${SIZES:-240 256}; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/r03_pmclen_${n}_$i
    rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r03_pmclen_${n}_$i --output-format csv -- python3 $R/bench.py --size $n --steps 10 --warmup 2 --no-cpu --no-512 --profile-steps 1 > $R/gpurun_out/r03_pmclen_${n}_$i.log 2>&1 || { tail -5 $R/gpurun_out/r03_pmclen_${n}_$i.log; echo "pass $i at $n failed"; }
  done
done
cd $R && python3 tools/pmc_len_table.py ${SIZES:-240 256}
