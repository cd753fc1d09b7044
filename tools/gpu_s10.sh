#!/bin/bash
# round-5 GPU session 10: 4-row tiles, units a wave takes in a row (RWKV_TILE_RUNMAX 1 / 2 / 4) at 14B and 1B5
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))"
}
{
echo "# 256 timed greedy steps, one box, max_ctx 1"
for m in 14B 1B5; do
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --model $m"
one "$m row (default)" A=1
one "$m tile run 4" RWKV_TILE=15
one "$m tile run 2" RWKV_TILE=15 RWKV_LIB=$V/lib_run2.so
one "$m tile run 1" RWKV_TILE=15 RWKV_LIB=$V/lib_run1.so
done
} > $O/tile_run_ab.txt 2>&1; cat $O/tile_run_ab.txt
