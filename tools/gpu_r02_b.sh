#!/bin/bash
# round-2 GPU session B: GPU test suite with the streaming-driver kernels, SPLIT default check
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r02b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( unset RWKV_LIB; echo "== base (in-tree: NBUF=1 SPLIT=23)"; timeout 200 python bench.py --steps 256 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
" ) > $O/sweep.log 2>&1
STEPS=256 timeout 300 bash tools/sweep.sh run "s7=" >> $O/sweep.log 2>&1
cat $O/sweep.log
