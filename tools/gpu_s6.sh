#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
bash tools/gpu_s5.sh
timeout 600 python tools/tile_check.py 8 > $O/tile_check_all.txt 2>&1; cat $O/tile_check_all.txt
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 1 --long-prompt 0 --config2-steps 0"
one() {   # label, tree, env...
  local label=$1 tree=$2; shift 2
  ( cd $tree && env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
{
echo "# 7B, 256 steps, max_ctx 32: tile-form decode kernels (RWKV_TILE mask) against the row-form kernels"
one "row form, carry default" $R A=1
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "RWKV_TILE=15 (all four, no carry)" $R RWKV_TILE=15
one "RWKV_TILE=13 (att, ffn_rk, ffnv)" $R RWKV_TILE=13
one "row form, carry default" $R A=1
one "RWKV_TILE=15 (all four, no carry)" $R RWKV_TILE=15
} > $O/tile_all_ab.txt 2>&1; cat $O/tile_all_ab.txt
