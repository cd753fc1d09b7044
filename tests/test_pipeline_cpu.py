"""Layer-pipeline host logic (rwkv_cpp_accelerated_amd.pipeline) on CPU: world_size-2 and -3 `gloo`
process groups, the oracle as the stage backend.  The pipelined, multi-stream result must equal
plain single-process greedy decoding of each stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from rwkv_cpp_accelerated_amd import modelfile as mf, pipeline
import parity

L, D, SEED, STEPS = 4, 64, 314, 5


def test_partition_layers():
    assert pipeline.partition_layers(32, 1) == [(0, 32)]
    for n_layers, n_stages, n_embed in [(32, 2, 4096), (32, 4, 4096), (32, 8, 4096), (40, 8, 5120), (12, 3, 768), (4, 4, 64)]:
        parts = pipeline.partition_layers(n_layers, n_stages, n_embed)
        assert parts[0][0] == 0 and parts[-1][1] == n_layers and len(parts) == n_stages
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:])) and all(l1 > l0 for l0, l1 in parts)
        if n_stages > 1 and n_layers >= 4 * n_stages:      # the last stage carries the head: not more layers than the others
            assert parts[-1][1] - parts[-1][0] <= max(l1 - l0 for l0, l1 in parts[:-1])
    with pytest.raises(ValueError):
        pipeline.partition_layers(2, 3)


class OracleStage:
    """pipeline stage computed by the CPU oracle (test backend)"""

    def __init__(self, tensors, l0, l1, n_slots):
        import ctypes as C
        import oracle_lib
        self.C, self.o = C, oracle_lib.Oracle()
        self.o.L.oracle_stage_forward.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64,
                                                  C.POINTER(C.c_void_p), C.c_uint64, C.c_void_p]
        self.o.L.oracle_stage_forward.restype = C.c_int
        self.m = self.o.from_tensors(L, D, tensors)
        self.l0, self.l1 = l0, l1
        self.state = self.m.new_state(slots=n_slots)
        self.x = torch.zeros(D, dtype=torch.float64)
        self.logits = np.zeros(mf.VOCAB, np.float32)

    def forward(self, token, slot, want_pick):
        C = self.C
        xn = self.x.numpy()
        sp = (C.c_void_p * 5)(*[s.ctypes.data for s in self.state])
        rc = self.o.L.oracle_stage_forward(self.m.h, int(token), xn.ctypes.data, self.l0, self.l1, sp, slot,
                                           self.logits.ctypes.data if self.l1 == L else None)
        assert rc == 0
        return parity.argmax_ban0(self.logits) if want_pick else None


def _worker(rank, world, port, first_tokens, q, n_streams=None):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    l0, l1 = pipeline.partition_layers(L, world, D)[rank]
    st = OracleStage(mf.synthetic_tensors(L, D, seed=SEED), l0, l1, world)
    picks = pipeline.run_pipeline(st, dist, rank, world, first_tokens, STEPS, n_streams=n_streams)
    if rank == world - 1:
        q.put(picks)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_equals_single_process(world, oracle):
    first = [11, 222, 3333][:world]
    om = oracle.from_tensors(L, D, mf.synthetic_tensors(L, D, seed=SEED))
    want = np.zeros((world, STEPS), np.int64)
    for k, tk in enumerate(first):
        st = om.new_state()
        for i in range(STEPS):
            tk = parity.argmax_ban0(om.forward([tk], st)[0]); want[k, i] = tk
    om.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, first, q)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=240)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world,n_streams", [(3, 2)])
def test_fewer_streams_than_stages_in_flight(world, n_streams, oracle):
    """bench.py's `one_stream` leg (N > 1): only the first n_streams streams' slots of the schedule are filled -- n_streams = 1 is ONE
    stream through all the stages, the pipeline's single-stream latency.  The filled streams must decode exactly as alone; the
    others' rows stay empty and nobody waits for a hop that is never sent."""
    first = [11, 222, 3333][:world]
    om = oracle.from_tensors(L, D, mf.synthetic_tensors(L, D, seed=SEED))
    want = np.zeros((world, STEPS), np.int64)
    for k, tk in enumerate(first[:n_streams]):
        st = om.new_state()
        for i in range(STEPS):
            tk = parity.argmax_ban0(om.forward([tk], st)[0]); want[k, i] = tk
    om.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, first, q, n_streams)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=240)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(got, want)
