#!/bin/bash
# Copy the summaries tools/profile_all.sh left under gpurun_out/<workload>/ into profiles/<round>/ (tracked) and merge their
# traffic.json entries.  usage (repo root, after the gpurun call): tools/collect_profiles.sh r03 [workload ...]
R=$1; shift
LIST=${*:-head pretrain fp8 fp8_cached siglip force_ep force_ep_direct pretrain_ep competition8 competition}
mkdir -p profiles/$R
for w in $LIST; do
  d=gpurun_out/$w
  [ -f $d/bench_${w}_per_launch.txt ] || { echo "missing $w"; continue; }
  cp $d/bench_$w.json profiles/$R/bench_$w.json
  cp $d/bench_${w}_under_rocprof.json profiles/$R/bench_${w}_under_rocprof.json
  cp $d/bench_${w}_per_launch.txt profiles/$R/bench_${w}_per_launch.txt
  cp $d/bench_${w}_kernel_stats.csv profiles/$R/bench_${w}_kernel_stats.csv
  python3 - "$d/traffic.json" "profiles/$R/traffic.json" "$w" <<'PY'
import json, sys
src, dst, w = sys.argv[1:]
s = json.load(open(src))
try:
    d = json.load(open(dst))
except (OSError, ValueError):
    d = {}
d["_comment"] = s["_comment"]
d.setdefault("workloads", {})[w] = s["workloads"][w]
json.dump(d, open(dst, "w"), indent=1)
open(dst, "a").write("\n")
PY
done
ls profiles/$R
