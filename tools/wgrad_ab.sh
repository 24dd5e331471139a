#!/bin/bash
# Weight-gradient kernel at short reductions (BASELINE config 5: 128 experts x ~512 rows, fp32 gradients): the library against its
# twin without its global stores, and the per-section cycle stamps (make -C competesmoe_amd/csrc wgvar).
# usage (GPU box, repo root): tools/wgrad_ab.sh <tag> [gemm_bench flags]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for a in 0 2 9; do
  L=$ROOT/competesmoe_amd/lib/libcsmoe_hip_wg$a.so; [ $a = 0 ] && L=$ROOT/competesmoe_amd/lib/libcsmoe_hip.so
  [ -f $L ] || continue
  echo "== variant $a (0 library, 2 no fp32 stores, 9 stamps)"
  CSMOE_LIB=$L python3 tools/gemm_bench.py --which tn1,tn2 --iters $([ $a = 9 ] && echo 1 || echo 10) "$@" 2>&1 | grep -v amdgpu.ids
done > $OUT/wgrad_ab.txt 2>&1
cat $OUT/wgrad_ab.txt
