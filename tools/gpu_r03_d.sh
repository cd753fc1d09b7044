#!/bin/bash
# round-3 session d: fused resid+site launch of the chunk path -- tests, timing with the fusion on and off; decode after the error-word fix
cd "$(dirname "$0")/.."
O=gpurun_out/r03d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_prefill_gpu.py tests/test_pipeline_gpu.py -q --timeout 600 -rs -x 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_prefill.log; cat $O/pytest_prefill.log
for f in 0 1; do echo "== RWKV_SEQ_FUSE=$f"; RWKV_SEQ_FUSE=$f timeout 300 python tools/prefill_bench.py 2>/dev/null | tail -1 | cut -c1-330; done > $O/prefill_fuse.txt 2>&1; cat $O/prefill_fuse.txt
timeout 400 python bench.py --steps 256 --ref-steps 0 --no-cpu-baseline --config2-steps 0 2>/dev/null | tail -1 > $O/bench.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r03d/bench.json"))
print("decode tok/s", d["value"], {k: v["us"] for k, v in d["kernels"].items()})
print("prefill", d["prefill"]["ms_per_chunk"], d["prefill"]["tokens_per_s"], "long", d["prefill"]["long_prompt"]["tokens_per_s"], "batched", d["batched_decode"]["aggregate_tokens_per_s"], d["batched_decode"]["streams_96"]["aggregate_tokens_per_s"])
PY
