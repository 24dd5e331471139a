#!/bin/bash
# Every bench workload once (regression sweep); one line per mode.  usage (GPU box, repo root): tools/bench_sweep.sh
run() {
  python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
g = d['roofline']['all_grouped_gemm'] if d.get('roofline') else {}
print('%-60s %8.3f ms/step %10.0f tokens/s   GEMM %s TFLOP/s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], g.get('TFLOP/s')))" "$@"
}
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --skew
run --steps 20 --warmup 5 --block
run --steps 20 --warmup 5 --block-unfused
run --steps 5 --warmup 2 --competition --experts 8
run --steps 30 --warmup 5 --tokens 12800 --seq 2560 --d-model 1152 --d-ff 4304 --experts 4
run --steps 30 --warmup 5 --tokens 12800 --seq 2560 --d-model 1152 --d-ff 4304 --experts 4 --competition
run --steps 10 --warmup 3 --stack pretrain
run --steps 10 --warmup 3 --stack pretrain --block
run --steps 10 --warmup 3 --force-ep --ep-chunks 1
run --steps 10 --warmup 3 --force-ep --ep-chunks 2
run --steps 200 --warmup 5
