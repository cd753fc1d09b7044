#!/bin/bash
# round 3, session j: chunk path -- element-wise kernels with one round trip (k_seq_resid by quads, k_seq_wkv's state up front, arguments first)
cd "$(dirname "$0")/.."
O=gpurun_out/r03j; mkdir -p $O
export PYTHONUNBUFFERED=1
V=$PWD/rwkv-cpp-accelerated_amd/csrc/variants
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_prefill_gpu.py tests/test_ref_parity_gpu.py -q --timeout 900 -x 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -5 > $O/pytest.log; cat $O/pytest.log
for v in ${LIBS:-ao seq1 ao seq1}; do
  echo "== lib_$v" >> $O/prefill_ab.txt
  RWKV_LIB=$V/lib_$v.so timeout 300 python tools/prefill_bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(round(j['ms_per_chunk'], 4), 'ms per 32-token chunk', round(j['value']), 'tok/s')" >> $O/prefill_ab.txt
done
cat $O/prefill_ab.txt
