for m in "60 10" "60 30"; do set -- $m
 for rep in 1 2; do
  for v in 0 1; do
    echo -n "link $1 lat $2 tail_per_array=$v: "
    KW_SLAB_TAIL_PER_ARRAY=$v timeout -k 10 200 python tools/emulate_rank.py --grid 512 --ranks 8 --rank 1 --steps 20 --link-gbs $1 --latency-us $2 2>&1 | tail -1 | cut -c1-110
  done
 done
done
