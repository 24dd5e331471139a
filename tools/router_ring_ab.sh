#!/bin/bash
# x chunks in flight per wave of the one-pass router: 2 (the build), 3 and 4 (make -C competesmoe_amd/csrc rfring), standalone and in
# the headline step.  usage (GPU box, repo root): tools/router_ring_ab.sh
for v in "" rfr4 rfr5 "" rfr4 rfr5; do
  lib=$PWD/competesmoe_amd/lib/libcsmoe_hip${v:+_$v}.so
  CSMOE_LIB=$lib python tools/router_bench.py 2>/dev/null | sed -n 1p | cut -c1-90 | sed "s/^/standalone ${v:-ring3} /"
  CSMOE_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-step ${v:-ring3}', d['ms_per_step'], d['roofline']['hbm_kernels']['gate_select'])"
done
for v in rfr4 rfr5; do
  CSMOE_LIB=$PWD/competesmoe_amd/lib/libcsmoe_hip_$v.so python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "gate_select or router" -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/tests $v: /"
done
