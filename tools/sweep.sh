#!/bin/bash
# build tuning variants of the engine (one -D per variant) and bench each on the GPU box
# usage: tools/sweep.sh build|run  "NAME=-DFLAG=V ..."
set -e
cd "$(dirname "$0")/.."
CS=rwkv-cpp-accelerated_amd/csrc
mkdir -p $CS/variants gpurun_out
mode=$1; shift
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  if [ "$mode" = build ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $flags $CS/engine.hip -o $CS/variants/lib_$name.so &
  else
    echo "== $name ($flags)"
    RWKV_LIB=$PWD/$CS/variants/lib_$name.so timeout 300 python bench.py --steps ${STEPS:-128} --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --model ${MODEL:-7B} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
"
  fi
done
wait
