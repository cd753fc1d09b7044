#!/bin/bash
# round-2 engine (with the one-launch token) on the small models: is the persistent kernel worth keeping for S <= 2?
cd "$(dirname "$0")/../rwkv-cpp-accelerated_amd/csrc/variants/r02tree"
O=../../../../gpurun_out/r03f; mkdir -p $O
for m in 169M 1B5 3B; do for mega in 0 1; do echo "== $m RWKV_MEGA=$mega"; RWKV_MEGA=$mega timeout 300 python bench.py --model $m --steps 256 --warmup 16 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  tok/s %.1f  ms/step %.4f  e2e %.3f' % (d['value'], d['ms_per_step'], d['end_to_end']['frac_of_8TBps']))"; done; done > $O/mega_small.txt 2>&1
cat $O/mega_small.txt
