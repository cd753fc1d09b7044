#!/bin/bash
# one-launch token: agreement with the launch kernels, engine parity tests, timeline, short bench (mega on / off)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/mega
{
  echo "######## mega_check"; timeout 600 python tools/mega_check.py 2>&1 | tail -12
  if [ -z "$NO_TESTS" ]; then echo "######## tests"; timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -5; fi
  for m in ${MODELS:-7B}; do
    echo "######## timeline $m"; timeout 300 python tools/mega_timeline.py $m 2>&1 | tail -22
    for mg in 1 0; do
      echo "######## bench $m RWKV_MEGA=$mg"
      RWKV_MEGA=$mg timeout 300 python bench.py --steps ${STEPS:-256} --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --model $m 2>gpurun_out/mega/bench_$m_$mg.err | tail -1 > gpurun_out/mega/bench_${m}_$mg.json
      python -c "
import json,sys
d=json.loads(open('gpurun_out/mega/bench_${m}_$mg.json').read())
print('  tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
" 2>&1 | tail -2
    done
  done
} 2>&1 | tee gpurun_out/mega/log.txt
