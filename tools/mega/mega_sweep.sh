#!/bin/bash
# one-launch token tuning: timeline + short bench per engine variant (built by tools/sweep.sh build "name=-D...")
# usage: tools/mega_sweep.sh name ...      ("base" = the in-tree library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/mega
for v in "${@:-base}"; do
  echo "######## $v"
  if [ "$v" = base ]; then unset RWKV_LIB; else export RWKV_LIB=$PWD/rwkv-cpp-accelerated_amd/csrc/variants/lib_$v.so; fi
  [ -n "$CHECK" ] && timeout 600 python tools/mega_check.py ${CHECK_SHAPES:-2 4096 2 2048} 2>&1 | tail -3
  for m in ${MODELS:-7B}; do
    [ -z "$NO_TL" ] && timeout 300 python tools/mega_timeline.py $m ${TL_LAYERS:-8} 2>&1 | tail -9
    timeout 300 python bench.py --steps ${STEPS:-128} --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --model $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $m tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
"
  done
done 2>&1 | tee -a gpurun_out/mega/sweep.txt
