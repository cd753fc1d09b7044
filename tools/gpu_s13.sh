#!/bin/bash
# round-5 GPU session 13: the RCCL stand-in with device-side waits (no blocking host function): pipeline tests three times over, the N = 2 dry run
cd "$(dirname "$0")/.."
R=$PWD; O=gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu --timeout 500 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15; done > $O/pipe_tests_s13.log 2>&1; tail -40 $O/pipe_tests_s13.log
RWKV_BENCH_BACKEND=gloo RWKV_BENCH_ONE_DEVICE=1 RWKV_RCCL_LIB=$R/tests/_build/libfake_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 64 --warmup 8 --prefill-chunks 2 2>$O/bench_n2_dryrun_native.err | grep '^{"metric"' | tail -1 > $O/bench_n2_dryrun_native.json; echo "native dry run: exit $? $(wc -c < $O/bench_n2_dryrun_native.json) bytes"
python - <<P
import json
d=json.load(open('$O/bench_n2_dryrun_native.json'))
print(d['value'], d['n_gpus'], d['config'].get('workload'), (d.get('two_streams_per_stage') or {}))
P
tail -5 $O/bench_n2_dryrun_native.err
