#!/bin/bash
# round-5 GPU session 11: 14B, 4-row tiles with one unit per wave turn: which classes gain over the row form
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --model 14B"
one() {   # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))"
}
{
echo "# 14B, 256 timed greedy steps, one box, max_ctx 1; tile kernels built with RWKV_TILE_RUNMAX=1"
one "row (default)" A=1
one "mask 4 (ffn_rk)" RWKV_TILE=4 RWKV_LIB=$V/lib_run1.so
one "mask 12 (ffn_rk, ffn_v)" RWKV_TILE=12 RWKV_LIB=$V/lib_run1.so
one "mask 5 (att, ffn_rk)" RWKV_TILE=5 RWKV_LIB=$V/lib_run1.so
one "row (default)" A=1
one "mask 4 (ffn_rk)" RWKV_TILE=4 RWKV_LIB=$V/lib_run1.so
one "mask 15" RWKV_TILE=15 RWKV_LIB=$V/lib_run1.so
} > $O/tile_14B_masks.txt 2>&1; cat $O/tile_14B_masks.txt
