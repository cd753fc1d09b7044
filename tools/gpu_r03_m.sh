#!/bin/bash
# round 3, session m: ring / carry knobs at the other model sizes with the final kernels.  usage: RUNS="model:ring:carry ..." tools/gpu_r03_m.sh
cd "$(dirname "$0")/.."
O=gpurun_out/r03m; mkdir -p $O
export PYTHONUNBUFFERED=1
for r in ${RUNS:-3B:13:0 3B:13:16 3B:13:12 3B:13:20 3B:13:0 3B:13:16 3B:13:12 3B:13:20}; do
  IFS=: read m ring carry <<< "$r"
  echo "== $m RWKV_RING=$ring RWKV_CARRY=$carry" >> $O/knobs.txt
  RWKV_RING=$ring RWKV_CARRY=$carry timeout 400 python bench.py --model $m --steps 512 --warmup 32 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --long-prompt 0 2>/dev/null | python tools/benchline.py >> $O/knobs.txt
done
cat $O/knobs.txt
