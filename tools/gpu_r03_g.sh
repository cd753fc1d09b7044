#!/bin/bash
# round-3 session g: att_out + ffn r/k as one launch (RWKV_FUSE=1): parity tests, A/B, timeline
cd "$(dirname "$0")/.."
O=gpurun_out/r03g; mkdir -p $O
export PYTHONUNBUFFERED=1
RWKV_FUSE=1 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_ref_parity_gpu.py -q --timeout 600 -x -k "not chunk_path" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -12 > $O/pytest_fuse.log; cat $O/pytest_fuse.log
for rep in 1 2; do for f in 0 1; do echo "== RWKV_FUSE=$f"; RWKV_FUSE=$f timeout 300 python bench.py --steps 256 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f' % (d['value'], d['ms_per_step'])); print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))"; done; done > $O/ab_fuse.txt 2>&1; cat $O/ab_fuse.txt
RWKV_FUSE=1 RWKV_TL_CLASS=3 timeout 150 python tools/timeline.py 7B 2>&1 | tail -24 > $O/timeline_fuse.txt; cat $O/timeline_fuse.txt
