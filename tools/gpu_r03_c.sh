#!/bin/bash
# round-3 session c: A/B of the error-word change on one box (interleaved, twice), then the new tests
cd "$(dirname "$0")/.."
O=gpurun_out/r03c; mkdir -p $O
export PYTHONUNBUFFERED=1
(for rep in 1 2; do STEPS=256 bash tools/sweep.sh run "base=x" "noherr=x"; done) > $O/ab_herr.txt 2>&1
cat $O/ab_herr.txt
timeout 1200 python -m pytest tests/test_dropin_gpu.py tests/test_pipeline_gpu.py -q --timeout 600 -rs 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_new1.log
timeout 1200 python -m pytest tests/test_prefill_gpu.py tests/test_ref_parity_gpu.py tests/test_engine_gpu.py -q --timeout 600 -rs -k "stages or parralel or chunk_path or abi" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_new2.log
cat $O/pytest_new1.log $O/pytest_new2.log
