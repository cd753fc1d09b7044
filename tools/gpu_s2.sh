#!/bin/bash
# round-5 GPU session 2: (a) where round 4 lost 2.7 % against the round-3 binary (same-box variants), (b) tile-form k_ffn_rk: parity + timing
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
V=$R/rwkv-cpp-accelerated_amd/csrc/variants
F="--steps 512 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0"
one() {   # label, tree, env...
  local label=$1 tree=$2; shift 2
  ( cd $tree && env "$@" timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-34s %.2f tok/s  ' % ('$label', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('att_kvr_wkv','att_out','ffn_rk','ffn_v','head') if n in k))" )
}
{
echo "# 7B, 512 timed greedy steps, one box, max_ctx 1 (no chunk path): which round-4 change costs what"
one "r03" $R/ab_r03 A=1
one "HEAD" $R A=1
one "HEAD RWKV_CARRY_COUNT=0" $R RWKV_CARRY_COUNT=0
one "HEAD -DRWKV_CARRY_VERIFY=0" $R RWKV_LIB=$V/lib_nover.so
one "HEAD verify, nobody waits" $R RWKV_LIB=$V/lib_nowait.so
one "HEAD nover + COUNT=0" $R RWKV_LIB=$V/lib_nover.so RWKV_CARRY_COUNT=0
one "r03 RWKV_CARRY=0" $R/ab_r03 RWKV_CARRY=0
one "HEAD RWKV_CARRY=0" $R RWKV_CARRY=0
one "r03" $R/ab_r03 A=1
one "HEAD" $R A=1
} > $O/r04_regression_bisect.txt 2>&1; cat $O/r04_regression_bisect.txt
timeout 300 python tools/tile_check.py 8 > $O/tile_check.txt 2>&1; cat $O/tile_check.txt
F="--steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 1 --long-prompt 0 --config2-steps 0"
{
echo "# 7B, 256 steps, max_ctx 32 (chunk path loaded: the tile kernel streams ITS image): tile-form k_ffn_rk against the row-form ring kernel"
one "row form, carry default" $R A=1
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "tile form k_ffn_rk (carry off)" $R RWKV_TILE=4
one "row form, RWKV_CARRY=0" $R RWKV_CARRY=0
one "tile form k_ffn_rk (carry off)" $R RWKV_TILE=4
} > $O/tile_frk_ab.txt 2>&1; cat $O/tile_frk_ab.txt
