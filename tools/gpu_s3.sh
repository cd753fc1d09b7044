#!/bin/bash
# round-5 GPU session 3: the carry's checksums requested at kernel entry (fix of round 4's -2.7 %), tile-form k_ffn_rk with the pair loader
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
F="--steps 512 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # label, tree, env...
  local label=$1 tree=$2; shift 2
  ( cd $tree && env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
{
echo "# 7B, 512 timed greedy steps, one box, max_ctx 1: HEAD with the carried rows' checksums requested at kernel entry"
one "r03" $R/ab_r03 A=1
one "HEAD (want at entry)" $R A=1
one "r03" $R/ab_r03 A=1
one "HEAD (want at entry)" $R A=1
one "HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
F="$F --model 3B"
one "3B r03" $R/ab_r03 A=1
one "3B HEAD (want at entry)" $R A=1
} > $O/carry_want_ab.txt 2>&1; cat $O/carry_want_ab.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -q --timeout 600 -k "carried or two_contexts" 2>&1 | tail -4
timeout 300 python tools/tile_check.py 8 > $O/tile_check2.txt 2>&1; cat $O/tile_check2.txt
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 1 --long-prompt 0 --config2-steps 0"
{
echo "# 7B, 256 steps, max_ctx 32: tile-form k_ffn_rk (pair loader) against the row-form ring kernel"
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "tile form k_ffn_rk (carry off)" $R RWKV_TILE=4
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "tile form k_ffn_rk (carry off)" $R RWKV_TILE=4
one "row form, carry default" $R A=1
} > $O/tile_frk_ab2.txt 2>&1; cat $O/tile_frk_ab2.txt
