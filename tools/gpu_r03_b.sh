#!/bin/bash
# round-3 session b: the new parity / boundary / transport tests, a short bench with the new legs, decode knob sweep
cd "$(dirname "$0")/.."
O=gpurun_out/r03b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_pipeline_gpu.py -q -x --timeout 600 -rs 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -15 > $O/pytest_new1.log
timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_ref_parity_gpu.py tests/test_engine_gpu.py -q --timeout 600 -rs -k "stages or parralel or chunk_path or abi" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -15 > $O/pytest_new2.log
cat $O/pytest_new1.log $O/pytest_new2.log
timeout 600 python bench.py --steps 256 --ref-steps 64 --cpu-seconds 5 2>$O/bench_short.err | tail -1 > $O/bench_short.json; cut -c1-400 $O/bench_short.json; tail -3 $O/bench_short.err
(for ring in 13 15 29 31; do echo "######## RWKV_RING=$ring"; RWKV_RING=$ring timeout 200 python bench.py --steps 128 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f' % (d['value'], d['ms_per_step'])); print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))"; done
STEPS=128 bash tools/sweep.sh run "prio1=x" "prio3=x") > $O/decode_knobs.txt 2>&1
cat $O/decode_knobs.txt
