for c in 0 1 2 4 8; do
  if [ $c = 0 ]; then F=""; else F="--force-ep --ep-chunks $c"; fi
  timeout -k 10 200 python bench.py --experts 8 $F --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ep8_$c.json 2> gpurun_out/ep8_$c.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/ep8_$c.json').read().strip().splitlines()[-1]); print('chunks $c', d['ms_per_step'])"
done
