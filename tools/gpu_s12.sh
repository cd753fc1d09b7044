#!/bin/bash
# round-5 GPU session 12: per-class residency (RWKV_TILE mask), 14B default = ffn k/r on 4-row tiles: tests, full-depth gate, bench
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -x -q -m gpu -k "each_decode_class or tile_form or drop_group or two_contexts or full_depth_gate or chunk_gate or long_prompt_vs_reference" 2>&1 | tail -8 > $O/pytest_s12.log; cat $O/pytest_s12.log
for m in 14B 7B; do
timeout 600 python bench.py --model $m --steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 2>$O/bench_${m}_s12.err | tail -1 > $O/bench_${m}_s12.json
python - <<P
import json
d=json.load(open('$O/bench_${m}_s12.json')); k=d['kernels']
print('$m', d['value'], d['end_to_end'], d['decode_form'], d['hbm_resident_bytes']['total'], {n:k[n]['us'] for n in k})
p=d.get('prefill') or {}
print('  prefill', p.get('ms_per_chunk'), (p.get('long_prompt') or {}).get('tokens_per_s'), (d.get('batched_decode') or {}).get('tokens_per_s'))
P
done
