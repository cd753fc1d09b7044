#!/bin/bash
# round-2 GPU session D: the rewritten chunked path -- parity tests, prefill bench + kernel statistics
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r02d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_prefill_gpu.py tests/test_pipeline_gpu.py tests/test_ref_parity_gpu.py tests/test_engine_gpu.py -m gpu -x -q --timeout 600 -k "not 14b and not full_depth" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -30 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/seq_trace.sh $O 2>&1 | tail -20
for M in 1B5 14B; do timeout 200 python tools/prefill_bench.py --model $M 2>/dev/null | tail -1 | cut -c1-330; done | tee $O/prefill_sizes.log
