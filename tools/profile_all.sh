#!/bin/bash
# Every workload DESIGN.md section 7 quotes, through tools/profile_round.sh (bench line, kernel trace, three PMC passes, per-launch
# table, traffic.json entry), one after the other.  usage (GPU box, repo root): tools/profile_all.sh [workload ...]
# Afterwards (here): tools/collect_profiles.sh r03 copies the summaries into profiles/r03/.
set -u
declare -A W=(
  [head]=""
  [pretrain]="--stack pretrain"
  [fp8]="--dtype fp8 --experts 128 --shared 2"
  [fp8_cached]="--dtype fp8 --experts 128 --shared 2 --fp8-weight-cache"
  [siglip]="--tokens 12800 --seq 2560 --d-model 1152 --d-ff 4304 --experts 4"
  [force_ep]="--force-ep"
  [force_ep_direct]="--force-ep --ep-direct"
  [pretrain_ep]="--stack pretrain --force-ep"
  [competition8]="--competition --experts 8"
  [competition]="--competition"
)
LIST=${*:-head pretrain fp8 fp8_cached siglip force_ep force_ep_direct pretrain_ep competition8 competition}
for w in $LIST; do
  echo "=== $w: bench.py ${W[$w]}  ($(date +%T))"
  st=20; [[ $w == competition* ]] && st=5
  STEPS=$st ROUND=3 tools/profile_round.sh $w ${W[$w]} > gpurun_out/profile_$w.log 2>&1 || { echo "FAILED $w"; tail -5 gpurun_out/profile_$w.log; }
  tail -3 gpurun_out/profile_$w.log | cut -c1-200
done
