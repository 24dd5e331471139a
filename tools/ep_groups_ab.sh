#!/bin/bash
# One rank's share of an 8-rank expert-parallel job on ONE GPU: 8 local experts of ~8 192 rows (bench.py --experts 8), without EP (0) and
# through the EP layer with 1 / 2 / 4 / 8 groups of local experts (the overlap depth): what the groups cost on the compute side.
# usage (GPU box, repo root): tools/ep_groups_ab.sh
for c in 0 1 2 4 8; do
  if [ $c = 0 ]; then F=""; else F="--force-ep --ep-chunks $c"; fi
  timeout -k 10 200 python bench.py --experts 8 $F --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ep8_$c.json 2> gpurun_out/ep8_$c.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/ep8_$c.json').read().strip().splitlines()[-1]); print('chunks $c', d['ms_per_step'])"
done
