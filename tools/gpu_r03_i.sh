#!/bin/bash
# round 3, session i: k_attout on the ring inside the carry chain (RWKV_RING=15) vs in register form (13)
cd "$(dirname "$0")/.."
O=gpurun_out/r03i; mkdir -p $O
export PYTHONUNBUFFERED=1
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
B="python bench.py --steps ${STEPS:-1024} --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --long-prompt 0"
for cfg in ${CFGS:-ao:13:32 ao:15:32 ao:13:32 ao:15:32 ao:15:16 ao:15:48}; do
  IFS=: read v r k <<< "$cfg"
  echo "== lib_$v RWKV_RING=$r RWKV_CARRY=$k ${MODEL:-7B}" >> $O/variants.txt
  RWKV_LIB=$V/lib_$v.so RWKV_RING=$r RWKV_CARRY=$k timeout 400 $B --model ${MODEL:-7B} 2>/dev/null | python tools/benchline.py >> $O/variants.txt
done
cat $O/variants.txt
[ -n "$SKIP_TESTS" ] || RWKV_RING=15 timeout 900 python -m pytest tests/test_engine_gpu.py -q --timeout 900 -x -k "greedy or parity or decode or state or forward" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -5
