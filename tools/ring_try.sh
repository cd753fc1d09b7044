#!/bin/bash
# LDS-ring decode kernels: engine parity tests with the ring on (RWKV_RING bitmask), then the decode bench per setting
# usage: [MODELS="1B5 14B"] [NO_TESTS=1] tools/ring_try.sh MASK ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ring
run_bench() {
  timeout 300 python bench.py --steps ${STEPS:-128} --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --model ${MODEL:-7B} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
"
}
[ $# -eq 0 ] && set -- 0 31
for mask in "$@"; do
  export RWKV_RING=$mask
  echo "######## RWKV_RING=$RWKV_RING"
  if [ "$RWKV_RING" != 0 ] && [ -z "$NO_TESTS" ]; then
    timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -3
  fi
  run_bench
  for m in ${MODELS:-}; do echo "-- $m"; MODEL=$m run_bench; done
done 2>&1 | tee -a gpurun_out/ring/log.txt
