#!/bin/bash
# One rank's share of an N-rank expert-parallel job on ONE GPU: E/N local experts (bench.py --experts E/N; 8 = a rank of the 8-GPU run,
# ~8 192 rows per expert), without EP (0) and through the EP layer with 1 / 2 / 4 / ... groups of local experts (the overlap depth):
# what the groups cost on the compute side.  usage (GPU box, repo root): tools/ep_groups_ab.sh [local experts = 8] [bench flags]
EL=${1:-8}; shift
TAG=ep${EL}$(echo "$*" | tr -d ' -')
for c in 0 1 2 4 8; do
  [ $c -gt $EL ] && continue
  if [ $c = 0 ]; then F=""; else F="--force-ep --ep-chunks $c"; fi
  timeout -k 10 200 python bench.py --experts $EL $F --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/${TAG}_$c.json 2> gpurun_out/${TAG}_$c.err || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_$c.json').read().strip().splitlines()[-1]); print('$EL local experts $*: groups $c', d['ms_per_step'], 'ms')"
done
