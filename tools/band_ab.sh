#!/bin/bash
# Row-space tile order: band height A/B on the headline step (libcsmoe_hip_band<B>.so from `make -C competesmoe_amd/csrc band B=<B>`;
# the library itself is band 4), every variant twice, interleaved.  usage (GPU box, repo root): [FLAGS='--competition --experts 8' STEPS=5] tools/band_ab.sh [4 5 6 8 ...]
LIST=${*:-4 3 5}
for rep in 1 2; do
  for b in $LIST; do
    v=_band$b; [ $b = 4 ] && v=""
    CSMOE_LIB=$PWD/competesmoe_amd/lib/libcsmoe_hip$v.so timeout -k 10 200 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline $FLAGS > gpurun_out/band_$b.json 2> gpurun_out/band_$b.err || exit 1
    python -c "
import json; d=json.loads(open('gpurun_out/band_$b.json').read().strip().splitlines()[-1]); k=d['kernels']; print('band $b', d['ms_per_step'], {n: round(v['ms'],3) for n, v in k.items() if 'gemm' in n or 'wgrad' in n})"
  done
done
