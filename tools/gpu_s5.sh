#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
RWKV_TL_CLASS=3 python tools/carry_tl.py 2>&1 | tail -2 | cut -c1-360 > $O/carry_verify_timeline_after.txt; cat $O/carry_verify_timeline_after.txt
F="--steps 512 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # label, tree, env...
  local label=$1 tree=$2; shift 2
  ( cd $tree && env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
{
echo "# 512 timed greedy steps, one box, max_ctx 1: HEAD = the carried rows' expected sums travel with the rows in LDS"
one "7B r03" $R/ab_r03 A=1
one "7B HEAD" $R A=1
one "7B r03" $R/ab_r03 A=1
one "7B HEAD" $R A=1
one "7B HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
F="$F --model 3B"
one "3B r03" $R/ab_r03 A=1
one "3B HEAD" $R A=1
one "3B HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
} > $O/carry_sums_in_lds_ab.txt 2>&1; cat $O/carry_sums_in_lds_ab.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -q --timeout 600 -k "carried or two_contexts or drop" 2>&1 | tail -4
