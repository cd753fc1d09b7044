#!/bin/bash
# round 3, session l: prefill bench over engine variants (RWKV_LIB) and RWKV_SEQ_PIPE masks, then GEMM timelines of one variant
cd "$(dirname "$0")/.."
O=gpurun_out/r03l; mkdir -p $O
export PYTHONUNBUFFERED=1
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
for cfg in ${CFGS:-p2nt:15 p2:15 p2nt:15 p2:15 p2:0}; do
  v=${cfg%%:*}; p=${cfg##*:}
  echo "== lib_$v RWKV_SEQ_PIPE=$p ${MODEL:-7B}" >> $O/ab.txt
  RWKV_LIB=$V/lib_$v.so RWKV_SEQ_PIPE=$p timeout 300 python tools/prefill_bench.py --model ${MODEL:-7B} 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(round(j['ms_per_chunk'], 4), 'ms per 32-token chunk', round(j['value']), 'tok/s')" >> $O/ab.txt
done
cat $O/ab.txt
if [ -n "$TL" ]; then for k in ${KINDS:-2 3 0 1}; do RWKV_LIB=$V/lib_$TL.so timeout 200 python tools/gemm_timeline.py $k 7B 2>&1 | grep -v "^loading\|^n_layers\|^n_embed\|amdgpu.ids" | tail -8; done > $O/gemm_timeline_$TL.txt 2>&1; cat $O/gemm_timeline_$TL.txt; fi
