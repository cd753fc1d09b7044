#!/bin/bash
# round-5 GPU session 9: the tile form generalised to 4-row tiles (14B: five per class and workgroup, 1B5: two): correctness against the row form, speed
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
for D in 4096 5120 2048; do echo "== D=$D"; timeout 300 python tools/tile_check.py 4 0 $D 2>&1 | grep "RWKV_TILE\|Error\|error" ; done > $O/tile_check_widths.txt 2>&1; cat $O/tile_check_widths.txt
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
one "7B tile (default)" A=1
one "7B row (RWKV_TILE=0)" RWKV_TILE=0
F="$F --model 14B"
one "14B row (default)" A=1
one "14B tile (RWKV_TILE=15)" RWKV_TILE=15
one "14B row (default)" A=1
one "14B tile (RWKV_TILE=15)" RWKV_TILE=15
one "14B tile 13 (attout in row form)" RWKV_TILE=13
F="${F/14B/1B5}"
one "1B5 row (default)" A=1
one "1B5 tile (RWKV_TILE=15)" RWKV_TILE=15
one "1B5 row (default)" A=1
one "1B5 tile (RWKV_TILE=15)" RWKV_TILE=15
} > $O/tile_widths_ab.txt 2>&1; cat $O/tile_widths_ab.txt
