#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2000 python -m pytest tests -m gpu -q --timeout 900 -rs -x 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py 2>$O/bench7b_full.err | tail -1 > $O/bench7b_full.json; echo "bench rc=${PIPESTATUS[0]}"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench7b_full.json'))
print(d['value'], d['ms_per_step'], d['end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['parity_gates_failed'])
print({k: (v['us'], round(v['GBps']/8000,3)) for k,v in d['kernels'].items()})
print(d['hbm_resident_bytes']['total'], d['load_s'], d['decode_form'])
print(d['parity_vs_reference_kernel'])
print(d['prefill']['ms_per_chunk'], d['prefill']['long_prompt']['tokens_per_s'], d['batched_decode']['ms_per_step'])
PY
