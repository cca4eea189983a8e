#!/bin/bash
# tools/emulate_rank.py over the slab schedules and the two transports, for a list of "grid ranks" cases and link models
#   CASES="512 8;256 8" MODELS="60 10;60 30;75 30" TRANSPORTS="p2p rccl" bash tools/emulate_schedules.sh
IFS=';' read -ra cases <<< "${CASES:-512 8;256 8;256 4;512 4}"
IFS=';' read -ra models <<< "${MODELS:-60 10}"
for m in "${models[@]}"; do set -- $m; gbs=$1; lat=$2
  for g in "${cases[@]}"; do set -- $g
    echo "== $1^3 / $2 ranks, link $gbs GB/s, latency $lat us"
    for tr in ${TRANSPORTS:-p2p rccl}; do
      for e in "slab_pipeline=0" "slab_batch=0,slab_chunks=1" "slab_batch=0,slab_chunks=2" "slab_batch=0,slab_chunks=4" "slab_batch=1" ""; do
        KW_TUNING="$e" timeout -k 10 200 python tools/emulate_rank.py --transport $tr --grid $1 --ranks $2 --rank 1 --steps 15 --link-gbs $gbs --latency-us $lat 2>&1 | tail -1 | cut -c1-220
      done
    done
  done
done
