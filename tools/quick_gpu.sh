#!/bin/bash
# quick GPU check of a tuning step: finite logits, timelines of the four per-layer kernels, short bench
# usage: tools/quick_gpu.sh [variant-name ...]   ("base" = the in-tree library)
# NB "base" is only a baseline when the in-tree library's stamp matches the working tree: bench.py rebuilds a stale one from the sources AS THEY ARE
# (on the GPU box too), i.e. from the very change under test.  Round 6 lost two A/Bs to that (profiles/r06/mfma_consumer_ab.txt): build BOTH sides as
# named variants, or check the line this script prints.
cd "$(dirname "$0")/.."
python - <<'PY'
import sys; sys.path.insert(0, ".")
from rwkv_cpp_accelerated_amd import build
import os
print("in-tree library:", "STALE against the working tree -- 'base' will be rebuilt from it" if build._stale(os.path.join(build.CSRC, "librwkv_mi355x.so"), build.engine_sources()) else "matches the working tree")
PY
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
