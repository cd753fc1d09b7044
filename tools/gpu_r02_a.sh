#!/bin/bash
# round-2 GPU session A: decode variants sweep, timeline, full bench with the reference-kernel gate, GPU test suite
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r02a; mkdir -p $O
export PYTHONUNBUFFERED=1
echo "== sweep 7B" | tee $O/sweep.log
( unset RWKV_LIB; echo "== base (in-tree)"; timeout 200 python bench.py --steps 256 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
" ) >> $O/sweep.log 2>&1
STEPS=256 timeout 600 bash tools/sweep.sh run "nb1=" "s15=" "s31=" "nb1s31=" >> $O/sweep.log 2>&1
echo "== sweep 1B5 / 14B (in-tree)" >> $O/sweep.log
for M in 1B5 14B; do ( unset RWKV_LIB; timeout 300 python bench.py --model $M --steps 256 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $M tok/s %.1f  ms/step %.4f  e2e %.4f of 8TB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['frac_of_8TBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
" ) >> $O/sweep.log 2>&1; done
cat $O/sweep.log
for k in 1 3 4; do echo "== timeline class $k"; RWKV_TL_CLASS=$k timeout 120 python tools/timeline.py 7B 2>&1 | head -10; done > $O/timeline.log 2>&1
cat $O/timeline.log
timeout 900 python bench.py --no-cpu-baseline 2>$O/bench7b.err | tail -1 > $O/bench7b.json; cut -c1-300 $O/bench7b.json; python - <<PY
import json
d=json.load(open("$O/bench7b.json"))
for k in ("ref_kernel_baseline","parity_vs_reference_kernel","prefill","sampled_decode","end_to_end","roofline"): print(k, d.get(k))
PY
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
