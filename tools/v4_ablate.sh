#!/bin/bash
# Where the one-wave-per-SIMD K-loop's time goes: the full loop against twins with its fragment reads / its LDS-DMA / both compiled
# out (make -C competesmoe_amd/csrc ablate), at one launch kind, plus the PMC of the full loops (v2 and v4).
# usage (GPU box, repo root): tools/v4_ablate.sh <tag> <which> [gemm_bench flags]
TAG=$1; WHICH=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for a in 0 1 2 3 4 5; do
  L=$ROOT/competesmoe_amd/lib/libcsmoe_hip_abl$a.so; [ $a = 0 ] && L=$ROOT/competesmoe_amd/lib/libcsmoe_hip.so
  echo "== ablation $a (1: no reads, 2: no DMA, 3: MFMAs + barriers only, 4: no landing wait, 5: no reads + no landing wait)"
  CSMOE_LIB=$L CSMOE_GEMM_KERNEL=v4 python3 tools/gemm_bench.py --which $WHICH --iters 10 "$@" 2>&1 | grep -v amdgpu.ids
done > $OUT/ablate.txt 2>&1
cat $OUT/ablate.txt
