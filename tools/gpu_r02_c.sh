#!/bin/bash
# round-2 GPU session C: pipeline tests (virtual-stage pipelined prefill, 14B split, native RCCL transport), chunk path
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_pipeline_gpu.py tests/test_prefill_gpu.py tests/test_engine_gpu.py tests/test_cpp_api.py -m gpu -q --timeout 900 -rs 2>&1 | grep -v "^loading\|^n_layers\|^n_embed" | tail -40 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
# dry run of bench.py --gpus 2 on the one GPU (gloo transport, Python schedule) to keep that path alive
RWKV_BENCH_BACKEND=gloo RWKV_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --model 1B5 2>&1 | tail -3 | cut -c1-600 > $O/bench_gpus2_dry.log; cat $O/bench_gpus2_dry.log
