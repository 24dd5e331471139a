#!/bin/bash
# Evidence for profiles/: one kernel-trace run and three SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the same
# bench command, then the per-launch summary and this workload's entry of gpurun_out/<tag>/traffic.json (copy both into profiles/rNN/
# TOGETHER: bench.py reads roofline.traffic from there).  usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [bench flags]
set -u
TAG=${1:-v6}
shift
EXTRA="$*"          # extra bench.py flags (e.g. --dtype fp8 --experts 128 --shared 2)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA"
python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline $EXTRA > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
# ONLY_BENCH=1: just the bench line again (after traffic.json of THIS build has been collected into profiles/rNN/, so that the committed
# line carries roofline.traffic instead of "collected on other kernel sources")
[ -n "${ONLY_BENCH:-}" ] && exit 0
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- $CMD > $OUT/bench_${TAG}_under_rocprof.json 2> $OUT/kt.err || exit 1
timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o t -- $CMD > /dev/null 2> $OUT/pf.err || exit 1
timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -o t -- $CMD > /dev/null 2> $OUT/pw.err || exit 1
timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pm -o t -- $CMD > /dev/null 2> $OUT/pm.err || exit 1
cd $ROOT
find $OUT -name "*.csv" | head -20
python3 tools/summarize_profile.py $(find $OUT/kt -name "*kernel_trace.csv") $OUT/bench_${TAG}_per_launch.txt \
  --fetch $(find $OUT/pf -name "*counter_collection.csv") --write $(find $OUT/pw -name "*counter_collection.csv") \
  --mfma $(find $OUT/pm -name "*counter_collection.csv") \
  --traffic-json $OUT/traffic.json --workload $TAG --bench-json $OUT/bench_${TAG}_under_rocprof.json \
  --title "round ${ROUND:-3}, build $TAG: rocprofv3 of \`python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA\` on MI355X" > /dev/null
cp $(find $OUT/kt -name "*kernel_stats.csv") $OUT/bench_${TAG}_kernel_stats.csv
# the counter CSVs are large: keep the summaries only
rm -rf $OUT/pf $OUT/pw $OUT/pm $OUT/kt/*/*kernel_trace.csv 2>/dev/null
head -30 $OUT/bench_${TAG}_per_launch.txt
