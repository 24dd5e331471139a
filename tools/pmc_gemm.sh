#!/bin/bash
# PMC of one grouped-GEMM launch kind for kernel v2 vs v4 (usage: tools/pmc_gemm.sh <tag> <which> [gemm_bench flags])
set -u
TAG=$1; WHICH=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in v2 v4; do
  export CSMOE_GEMM_KERNEL=$k
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$k -o t -- python3 $ROOT/tools/gemm_bench.py --which $WHICH --iters 4 "$@" > $OUT/kt_$k.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pm_$k -o t -- python3 $ROOT/tools/gemm_bench.py --which $WHICH --iters 4 "$@" > $OUT/pm_$k.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pl_$k -o t -- python3 $ROOT/tools/gemm_bench.py --which $WHICH --iters 4 "$@" > $OUT/pl_$k.log 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for k in ("v2", "v4"):
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/kt_{k}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gg" in r["Kernel_Name"]:
                dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pm", "pl"):
        for f in glob.glob(f"{out}/{d}_{k}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gg" in r["Kernel_Name"]:
                    cnt[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, v in dur.items():
        c = {n: sum(x) / len(x) for n, x in cnt[name].items()}
        ms = sum(v) / len(v)
        line = f"{k} {name[:40]:40s} {ms:7.3f} ms"
        if "GRBM_GUI_ACTIVE" in c:
            line += f" mfma_util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} clk {c['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9:.2f}"
        for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"):
            if n in c:
                line += f" {n[3:]}={c[n]:.3g}"
        if "TCC_HIT_sum" in c:
            line += f" L2hit {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}"
        print(line)
PY
rm -rf $OUT/kt_* $OUT/pm_v* $OUT/pl_v*/ 2>/dev/null
