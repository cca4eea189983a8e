for n in ${SIZES:-192 240 320 384}; do
  for g in "" "--granular"; do
    echo "== $n $g" >> gpurun_out/sizes.log
    timeout -k 10 200 python bench.py --size $n --no-cpu --steps 50 --warmup 5 $g 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('step'))" >> gpurun_out/sizes.log 2>&1 || exit 1
  done
done
