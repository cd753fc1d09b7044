#!/bin/bash
# quick GPU check of a tuning step: finite logits, timelines of the four per-layer kernels, short bench
# usage: tools/quick_gpu.sh [variant-name ...]   ("base" = the in-tree library)
cd "$(dirname "$0")/.."
for v in "${@:-base}"; do
  echo "######## $v"
  if [ "$v" = base ]; then unset RWKV_LIB; else export RWKV_LIB=$PWD/rwkv-cpp-accelerated_amd/csrc/variants/lib_$v.so; fi
  timeout 100 python tools/debug_nan.py 2>&1 | tail -1
  if [ -z "$NO_TL" ]; then for k in 1 2 3 4; do echo "== class $k"; RWKV_TL_CLASS=$k timeout 120 python tools/timeline.py ${MODEL:-7B} 2>&1 | tail -8 | awk '{printf "%s | ", $0} END {print ""}' | sed 's/  */ /g'; done; fi
  timeout 200 python bench.py --steps ${STEPS:-128} --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --model ${MODEL:-7B} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tok/s %.1f  ms/step %.4f' % (d['value'], d['ms_per_step'])); print({k: round(v['us'],2) for k,v in d['kernels'].items()})"
done
