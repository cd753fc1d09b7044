cd /root/repo
for v in ${VARIANTS:-base}; do
  echo "######## $v"
  if [ "$v" = base ]; then unset RWKV_LIB; else export RWKV_LIB=$PWD/rwkv-cpp-accelerated_amd/csrc/variants/lib_$v.so; fi
  for m in ${MODELS:-7B 14B}; do
  MODEL=$m timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline --ref-steps 0 --prefill-chunks 0 --config2-steps 0 --model $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  $m tok/s %.1f  ms/step %.4f  e2e %.0f GB/s' % (d['value'], d['ms_per_step'], d['end_to_end']['achieved_GBps']))
print('  ' + '  '.join('%s %.2f' % (k, v['us']) for k, v in d['kernels'].items()))
"
  done
done
