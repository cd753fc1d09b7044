#!/bin/bash
# round-5 GPU session 14: the N = 4 and N = 8 bench lines dry on the one GPU (ranks share cuda:0; gloo control plane; the engine's native schedule over the RCCL stand-in)
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
for N in 4 8; do
  RWKV_BENCH_BACKEND=gloo RWKV_BENCH_ONE_DEVICE=1 RWKV_RCCL_LIB=$R/tests/_build/libfake_rccl.so RWKV_PIPE_LOG=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29520 + N)) bench.py --gpus $N --steps 32 --warmup 4 --prefill-chunks 2 2>$O/bench_n${N}_dryrun_native.err | grep '^{"metric"' | tail -1 > $O/bench_n${N}_dryrun_native.json; echo "N=$N native dry run: exit $? $(wc -c < $O/bench_n${N}_dryrun_native.json) bytes"
  python - <<P
import json
try:
    d=json.load(open('$O/bench_n${N}_dryrun_native.json'))
    print(d['value'], d['n_gpus'], d['ms_per_step'], d['scaling'], d['config'].get('workload'))
    print(' dual', d.get('two_streams_per_stage'))
    print(' prefill', {k: v for k, v in (d.get('prefill') or {}).items() if k in ('tokens_per_s', 'ms', 'prompt_tokens', 'error')})
    print(' keys', [k for k in d if k not in ('config',)][:40])
except Exception as e:
    print('no line', e)
P
  grep -v "amdgpu.ids\|socket.cpp\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/bench_n${N}_dryrun_native.err | tail -8
done
