#!/bin/bash
# Pretrain stack with / without the single cast of x (CSMOE_OPERAND_FORK), alternating on one box.  usage (GPU box, repo root):
# tools/operand_fork_ab.sh
for f in 0 1 0 1; do
  for w in "--stack pretrain" "--dtype fp8 --experts 128 --shared 2 --fp8-weight-cache" "--stack pretrain --force-ep"; do
    CSMOE_OPERAND_FORK=$f python bench.py $w --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CSMOE_OPERAND_FORK=$f', '$w', d['ms_per_step'], 'ms/step')"
  done
done
