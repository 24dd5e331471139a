#!/bin/bash
# What the vendor's dense GEMM kernel IS at the headline shapes: kernel name (macro tile, depth-U, LDS
# layout are encoded in it), duration, FETCH_SIZE, MFMA busy and clock, next to the grouped kernels' in profiles/.
# usage (GPU box, repo root): tools/vendor_probe.sh <tag>
set -u
TAG=${1:-vendor}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 tools/lib_gemm_probe.py > $OUT/probe.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/lib_gemm_probe.py --few"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- $CMD > /dev/null 2> $OUT/kt.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o t -- $CMD > /dev/null 2> $OUT/pf.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pm -o t -- $CMD > /dev/null 2> $OUT/pm.err || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pl -o t -- $CMD > /dev/null 2> $OUT/pl.err || exit 1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def load(pat):
    f = glob.glob(pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
tr = load(out + "/kt/**/*kernel_trace.csv")
d = collections.defaultdict(list)
meta = {}
for r in tr:
    k = (r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    meta[k] = (r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"))
def pmc(pat, names):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in load(pat):
        if r["Counter_Name"] in names:
            a[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in a.items()}
pf = pmc(out + "/pf/**/*counter_collection.csv", {"FETCH_SIZE"})
pm = pmc(out + "/pm/**/*counter_collection.csv", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
pl = pmc(out + "/pl/**/*counter_collection.csv", {"TCC_HIT_sum", "TCC_MISS_sum"})
with open(out + "/vendor_kernels.txt", "w") as fo:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 1.0:
            continue
        name, gs, wg = k
        avg = sum(v) / len(v)
        f = pf.get((name, gs), {}).get("FETCH_SIZE")
        m = pm.get((name, gs), {})
        l = pl.get((name, gs), {})
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m else float("nan")
        clk = m["GRBM_GUI_ACTIVE"] / 8 / (avg * 1e-3) / 1e9 if m else float("nan")
        hit = l["TCC_HIT_sum"] / (l["TCC_HIT_sum"] + l["TCC_MISS_sum"]) if l else float("nan")
        fo.write(f"{avg:8.3f} ms x{len(v):3d}  grid {gs:>9s} wg {wg:>4s}  lds/vgpr/agpr/sgpr {meta[k]}  FETCH {f * 1024 / 1e9 if f else float('nan'):7.2f} GB  "
                 f"mfma_util {util:.3f} clk {clk:.2f} GHz  L2 hit {hit:.3f}\n    {name[:400]}\n")
print(open(out + "/vendor_kernels.txt").read())
PY
rm -rf $OUT/pf $OUT/pm $OUT/pl $OUT/kt
