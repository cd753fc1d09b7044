#!/bin/bash
# round 3, session k: the chunk path's GEMM as a software pipeline over k-blocks (k_seq_gemm_p; RWKV_SEQ_PIPE bit per GEMM kind)
cd "$(dirname "$0")/.."
O=gpurun_out/r03k; mkdir -p $O
export PYTHONUNBUFFERED=1
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_prefill_gpu.py tests/test_ref_parity_gpu.py tests/test_pipeline_gpu.py -q --timeout 900 -x 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -5 > $O/pytest.log; cat $O/pytest.log
for p in ${PIPES:-0 15 0 15 1 2 4 8}; do
  echo "== RWKV_SEQ_PIPE=$p ${MODEL:-7B}" >> $O/pipe_ab.txt
  RWKV_SEQ_PIPE=$p timeout 300 python tools/prefill_bench.py --model ${MODEL:-7B} 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(round(j['ms_per_chunk'], 4), 'ms per 32-token chunk', round(j['value']), 'tok/s')" >> $O/pipe_ab.txt
done
cat $O/pipe_ab.txt
for k in ${KINDS:-2 3 0 1}; do timeout 200 python tools/gemm_timeline.py $k 7B 2>&1 | grep -v "^loading\|^n_layers\|^n_embed\|amdgpu.ids" | tail -9; done > $O/gemm_timeline_p.txt 2>&1; cat $O/gemm_timeline_p.txt
