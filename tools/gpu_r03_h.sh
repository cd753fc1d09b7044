#!/bin/bash
# round 3, session h: carry with the check in the idle prologue waves
cd "$(dirname "$0")/.."
O=gpurun_out/r03h; mkdir -p $O
export PYTHONUNBUFFERED=1
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_engine_gpu.py -q --timeout 900 -x -k "carried or hand_off or greedy or parity or decode" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -12 > $O/pytest.log; cat $O/pytest.log
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
B="python bench.py --steps ${STEPS:-1024} --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --long-prompt 0"
for cfg in ${CFGS:-old:0 cur:32 nover:32 cur:0 old:0 cur:32 nover:32 cur:24 cur:40}; do
  v=${cfg%%:*}; k=${cfg##*:}
  echo "== lib_$v RWKV_CARRY=$k ${MODEL:-7B}" >> $O/variants.txt
  RWKV_LIB=$V/lib_$v.so RWKV_CARRY=$k timeout 400 $B --model ${MODEL:-7B} 2>/dev/null | python tools/benchline.py >> $O/variants.txt
done
cat $O/variants.txt
