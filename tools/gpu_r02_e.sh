#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_prefill_gpu.py tests/test_engine_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q --timeout 600 -k "not 14b" 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/seq_trace.sh $O 2>&1 | tail -16
