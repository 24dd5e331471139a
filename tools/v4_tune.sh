#!/bin/bash
# Slot-table variants of the one-wave-per-SIMD K-loop as twin libraries (plain epilogue only: -DCSMOE_V4_DEV), benched on the
# plain-epilogue launches.  usage: tools/v4_tune.sh build   (here, no GPU)   |   tools/v4_tune.sh run <tag> [gemm_bench flags]  (GPU box)
ROOT=$(cd $(dirname $0)/.. && pwd); CS=$ROOT/competesmoe_amd/csrc; LIB=$ROOT/competesmoe_amd/lib
VARIANTS=(
 "A:-DCSMOE_V4_RSTEP=2 -DCSMOE_V4_BAR=34 -DCSMOE_V4_D0=36 -DCSMOE_V4_DSTEP=5 -DCSMOE_V4_LAND=94 -DCSMOE_V4_R0=96 -DCSMOE_V4_R0STEP=2"
 "G:-DCSMOE_V4_RSTEP=2 -DCSMOE_V4_BAR=34 -DCSMOE_V4_D0=36 -DCSMOE_V4_DSTEP=5 -DCSMOE_V4_LAND=78 -DCSMOE_V4_R0=80 -DCSMOE_V4_R0STEP=3"
 "L:-DCSMOE_V4_RSTEP=2 -DCSMOE_V4_BAR=34 -DCSMOE_V4_D0=36 -DCSMOE_V4_DSTEP=5 -DCSMOE_V4_LAND=62 -DCSMOE_V4_R0=64 -DCSMOE_V4_R0STEP=4"
 "M:-DCSMOE_V4_RSTEP=2 -DCSMOE_V4_BAR=34 -DCSMOE_V4_D0=36 -DCSMOE_V4_DSTEP=5 -DCSMOE_V4_LAND=62 -DCSMOE_V4_R0=64 -DCSMOE_V4_R0STEP=2"
)
if [ "$1" = build ]; then
  mkdir -p $LIB/obj_tune
  OBJS=$(ls $LIB/obj/*.o | grep -v gemm_bf16_v4.o)
  for v in "${VARIANTS[@]}"; do
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -DCSMOE_V4_DEV $f -c $CS/gemm_bf16_v4.hip -o $LIB/obj_tune/v4_$n.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $LIB/obj_tune/v4_$n.o -o $LIB/libcsmoe_hip_tune$n.so || exit 1
    echo built $n
  done
  exit 0
fi
TAG=$2; shift 2
mkdir -p $ROOT/gpurun_out/$TAG
for v in "${VARIANTS[@]}"; do
  n=${v%%:*}
  echo "== variant $n: ${v#*:}"
  CSMOE_LIB=$LIB/libcsmoe_hip_tune$n.so CSMOE_GEMM_KERNEL=v4 python3 $ROOT/tools/gemm_bench.py --which nt2,nn2 --iters 10 "$@" 2>&1 | grep -v amdgpu.ids
done | tee $ROOT/gpurun_out/$TAG/tune.txt
