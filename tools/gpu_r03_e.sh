#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03e; mkdir -p $O
export PYTHONUNBUFFERED=1
(for rep in 1 2; do STEPS=256 bash tools/sweep.sh run "base=x" "noherr=x"; done) > $O/ab_herr2.txt 2>&1
cat $O/ab_herr2.txt
timeout 1200 python -m pytest tests/test_prefill_gpu.py tests/test_engine_gpu.py -q --timeout 600 -rs -x 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -8 > $O/pytest.log; cat $O/pytest.log
