#!/bin/bash
# round-5 GPU session 4: uniform role branches + clean-slate loader + checksums requested at entry; tile-form k_ffn_rk; other sizes
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
echo "# 512 timed greedy steps, one box, max_ctx 1: HEAD = uniform role branches, clean-slate loader, carried rows' checksums requested at kernel entry"
one "7B r03" $R/ab_r03 A=1
one "7B HEAD" $R A=1
one "7B r03" $R/ab_r03 A=1
one "7B HEAD" $R A=1
one "7B HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
F="$F --model 3B"
one "3B r03" $R/ab_r03 A=1
one "3B HEAD" $R A=1
F="${F/3B/14B}"
one "14B r03" $R/ab_r03 A=1
one "14B HEAD" $R A=1
F="${F/14B/1B5}"
one "1B5 r03" $R/ab_r03 A=1
one "1B5 HEAD" $R A=1
F="${F/1B5/169M}"
one "169M r03" $R/ab_r03 A=1
one "169M HEAD" $R A=1
} > $O/uniform_roles_ab.txt 2>&1; cat $O/uniform_roles_ab.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -q --timeout 600 -x 2>&1 | tail -4
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 1 --long-prompt 0 --config2-steps 0"
{
echo "# 7B, 256 steps, max_ctx 32: tile-form k_ffn_rk (pair loader) against the row-form ring kernel"
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "tile form k_ffn_rk (carry off)" $R RWKV_TILE=4
one "row form, carry default" $R A=1
} > $O/tile_frk_ab3.txt 2>&1; cat $O/tile_frk_ab3.txt
