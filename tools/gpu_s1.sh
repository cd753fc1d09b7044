#!/bin/bash
# round-5 GPU session 1: GPU tests, same-box A/B of the round-3 binary against HEAD, the full bench line, long-prompt stage sweep
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rs 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -25 > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
bash tools/ab_r03.sh 3 > $O/r03_vs_head_ab.txt 2>&1; cat $O/r03_vs_head_ab.txt
timeout 900 python bench.py 2>$O/bench7b_full.err | tail -1 > $O/bench7b_full.json; echo "bench rc=${PIPESTATUS[0]}"; cut -c1-400 $O/bench7b_full.json
for s in 2 3 4; do RWKV_SEQ_STAGES=$s timeout 200 python tools/long_prompt_bench.py 7B 512 2>/dev/null | tail -1; done > $O/stages_sweep.txt; cat $O/stages_sweep.txt
