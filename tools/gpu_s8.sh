#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
F="--steps 512 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # label, tree, env...
  local label=$1 tree=$2; shift 2
  ( cd $tree && env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
{
echo "# 512 timed greedy steps, one box, max_ctx 1: ROW form (RWKV_TILE=0 at 7B), carried rows checked by the wave that takes them against sums that travel in LDS"
one "7B r03" $R/ab_r03 A=1
one "7B HEAD row form" $R RWKV_TILE=0
one "7B r03" $R/ab_r03 A=1
one "7B HEAD row form" $R RWKV_TILE=0
one "7B HEAD row form RWKV_CARRY=0" $R RWKV_TILE=0 RWKV_CARRY=0
one "7B HEAD tile form (default)" $R A=1
F="$F --model 3B"
one "3B r03" $R/ab_r03 A=1
one "3B HEAD" $R A=1
one "3B r03" $R/ab_r03 A=1
one "3B HEAD" $R A=1
one "3B HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
} > $O/carry_check_at_take_ab.txt 2>&1; cat $O/carry_check_at_take_ab.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -q --timeout 600 -k "carried or two_contexts or hand_off or tile_form" 2>&1 | tail -4
timeout 300 python tools/tile_check.py 8 1 2>&1 | grep -v "^RWKV_TILE\|amdgpu.ids" > $O/tile_timeline_att.txt; cat $O/tile_timeline_att.txt
timeout 300 python tools/tile_check.py 8 4 2>&1 | grep -v "^RWKV_TILE\|amdgpu.ids" > $O/tile_timeline_ffnv.txt; cat $O/tile_timeline_ffnv.txt
