#!/bin/bash
# round-5 GPU session 15: workgroups per decode launch (RWKV_GRID) at the small widths: is 256 the right grid where a workgroup owns 3-10 channels?
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
one() {   # model grid
  RWKV_GRID=$2 timeout 200 python bench.py --model $1 --steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernels']
    print('%-5s grid %-4s %8.1f tok/s  ' % ('$1', '$2', d['value']) + '  '.join('%s %.2f' % (n, k[n]['us']) for n in ('first','att_kvr_wkv','att_out','ffn_rk','ffn_v','head','argmax') if n in k))
except Exception as e:
    print('$1 grid $2: no line', e)"
}
{
echo "# 256 timed greedy steps, one box, max_ctx 1"
for m in 169M 430M 1B5 3B; do
  for g in 256 192 128 96 64 48 32; do one $m $g; done
done
} > $O/grid_sweep.txt 2>&1; cat $O/grid_sweep.txt
