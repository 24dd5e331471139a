#!/bin/bash
# Weight-gradient stores: non-temporal (the library) against plain (libcsmoe_hip_wgnt0.so = gemm_bf16_v2p.hip with -DCSMOE_WG_NT=0),
# kernel alone at three shapes and the headline / pretrain / config-5 steps.  usage (GPU box, repo root): tools/wgrad_nt_ab.sh
R=$PWD
for rep in 1 2; do for v in "" _wgnt0; do
  export CSMOE_LIB=$R/competesmoe_amd/lib/libcsmoe_hip$v.so
  echo "== lib$v"
  python3 tools/gemm_bench.py --which tn1,tn2 --iters 10 2>&1 | grep "^tn"
  python3 tools/gemm_bench.py --which tn1,tn2 --iters 10 --f32-out 2>&1 | grep "^tn"
  python3 tools/gemm_bench.py --which tn1,tn2 --iters 10 --f32-out --E 128 2>&1 | grep "^tn"
  for f in "" "--stack pretrain" "--dtype fp8 --experts 128 --shared 2 --fp8-weight-cache"; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline $f > gpurun_out/nt.json 2> gpurun_out/nt.err || exit 1
    python -c "
import json; d=json.loads(open('gpurun_out/nt.json').read().strip().splitlines()[-1]); k=d['kernels']; print('step [$f]', d['ms_per_step'], {n: round(x['ms'],3) for n, x in k.items() if 'wgrad' in n})"
  done
done; done
