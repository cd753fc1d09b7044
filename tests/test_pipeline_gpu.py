"""Layer pipeline on the GPU box (BASELINE config 4 and SURVEY 8f row 2): pipelined prompt chunks across stages, the
14B-shaped stage split, and the engine's native RCCL transport (as far as one GPU allows)."""
import os
import socket
import sys

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from rwkv_cpp_accelerated_amd import engine
    engine.lib()
    return engine


@pytest.mark.parametrize("S,n_tokens", [(2, 256), (3, 77), (4, 130)])
def test_pipelined_prefill_over_virtual_stages(eng_mod, S, n_tokens):
    """RWKV::loadContext (rwkv.h:395-413) with the prompt's 32-token chunks as micro-batches over S stages co-located on one
    GPU (same code path as rwkv_pipe_prefill minus the RCCL hop: stage s works on chunk t - s at tick t, the chunk's residual
    stream handed from stage to stage): logits of the last position, every stage's share of the recurrent state and the
    continuation must equal the single-context run -- bit for bit (same kernels, same data)."""
    from rwkv_cpp_accelerated_amd import pipeline
    L, D = 6, 768
    t = mf.synthetic_tensors(L, D, seed=91)
    rng = np.random.default_rng(S)
    prompt = [int(x) for x in rng.integers(2, mf.VOCAB, n_tokens)]
    full = eng_mod.RWKV(resident=True); full.loadTensors(L, D, t, maxGPT=32)
    for i in range(0, n_tokens, 32):
        last_logits = full.forward(prompt[i:i + 32], eng_mod.MODE_GPT)[: len(prompt[i:i + 32]) * mf.VOCAB].reshape(-1, mf.VOCAB)[-1].copy()
    full.pull_state(1)
    parts = pipeline.partition_layers(L, S, D)
    stages = [pipeline.EngineStage(t, L, D, l0, l1, n_slots=1, prefill=True) for l0, l1 in parts]
    chunks = [prompt[i:i + 32] for i in range(0, n_tokens, 32)]
    for tick in range(len(chunks) + S - 1):
        for s in reversed(range(S)):              # any order that respects "stage s reads what stage s-1 produced last tick"
            ci = tick - s
            if not 0 <= ci < len(chunks):
                continue
            n = len(chunks[ci])
            if s > 0:
                stages[s].m.xseq_copy_from(stages[s - 1].m, n, buf=ci & 1)
            stages[s].m.stage_chunk(chunks[ci] if s == 0 else None, n, row0=0, buf=ci & 1)
        for st in stages:
            st.m.sync()
    got = stages[-1].m.logits(32)[: len(chunks[-1]) * mf.VOCAB].reshape(-1, mf.VOCAB)[-1]
    assert np.array_equal(got, last_logits)
    for (l0, l1), st in zip(parts, stages):
        st.m.pull_state(1)
        for a, b in zip(st.m.state.arrays(), full.state.arrays()):
            assert np.array_equal(a[l0 * D:l1 * D], b[l0 * D:l1 * D])
    # the decode that follows continues from the same state on both sides
    tk_f = tk_p = parity.argmax_ban0(last_logits)
    for _ in range(4):
        tk_f = full.stage_forward(tk_f, 0, want_pick=True)
        for i, st in enumerate(stages):
            if i > 0:
                st.x.copy_(stages[i - 1].x); st.torch.cuda.synchronize()
            nxt = st.forward(tk_p, 0, want_pick=st.last)
        tk_p = nxt
        assert tk_p == tk_f
    for st in stages:
        st.m.close()
    full.close()


@pytest.mark.parametrize("S", [2, 4, 8])
def test_14b_shape_virtual_stages(eng_mod, S):
    """BASELINE config 4's split (RWKV-4-14B: L=40, D=5120 over 2/4/8 stages) with all stages on this box's one GPU: the
    chained stage contexts reproduce the whole-model greedy ids and logits bit for bit"""
    import torch
    from rwkv_cpp_accelerated_amd import pipeline
    L, D = mf.SHAPES["14B"]
    t = mf.synthetic_tensors_torch(L, D, seed=14, device="cuda")
    torch.cuda.synchronize()
    full = eng_mod.RWKV(resident=True); full.loadTensors(L, D, t, maxGPT=1)
    parts = pipeline.partition_layers(L, S, D)
    assert parts[0][0] == 0 and parts[-1][1] == L and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    stages = [pipeline.EngineStage(t, L, D, l0, l1, n_slots=1) for l0, l1 in parts]
    del t
    tk_p = tk_f = 4242
    for step in range(5):
        tk_f = full.stage_forward(tk_f, 0, want_pick=True)
        for i, st in enumerate(stages):
            if i > 0:
                st.x.copy_(stages[i - 1].x); torch.cuda.synchronize()
            nxt = st.forward(tk_p, 0, want_pick=st.last)
        tk_p = nxt
        assert tk_p == tk_f, step
        assert np.array_equal(full.logits(1)[: mf.VOCAB], stages[-1].m.logits(1)[: mf.VOCAB])
    for st in stages:
        st.m.close()
    full.close()
    torch.cuda.empty_cache()


def test_native_transport_single_rank(eng_mod):
    """rwkv_pipe_init / rwkv_pipe_decode / rwkv_pipe_prefill with world = 1 on the REAL library: librccl.so is resolved, a
    communicator is made, and rwkv_pipe_init's agreement round makes its one hop to the rank itself -- a grouped ncclSend + ncclRecv of
    8 x uint64 on the engine's stream, checked byte for byte: the entry points, datatype codes, group and stream order the multi-GPU
    schedule is built on, executed by the library the driver's 8-GPU run will use.  Then the tick loop, the control-block ring and the
    device-side id feedback run (no further hop to make): must equal decode_greedy and the chunked forward"""
    L, D, n = 3, 768, 24
    t = mf.synthetic_tensors(L, D, seed=92)
    a = eng_mod.RWKV(resident=True); a.loadTensors(L, D, t, maxGPT=32)
    b = eng_mod.RWKV(resident=True); b.loadTensors(L, D, t, maxGPT=32)
    b.pipe_init(eng_mod.RWKV.pipe_unique_id(), 0, 1)
    info = b.pipe_info()
    assert info["world"] == 1 and info["rccl_version"] > 20000 and "fake" not in info["rccl_path"], info      # (2.x.y as 2xxyy: not the test stand-in)
    prompt = [int(x) for x in np.random.default_rng(2).integers(2, mf.VOCAB, 70)]
    for i in range(0, 70, 32):
        a.forward(prompt[i:i + 32], eng_mod.MODE_GPT)
    b.pipe_prefill(prompt, len(prompt))
    assert np.array_equal(a.logits(32)[: 6 * mf.VOCAB], b.logits(32)[: 6 * mf.VOCAB])
    want = a.decode_greedy(77, n)
    got = b.pipe_decode([77], n, 1, last=True)
    assert np.array_equal(got[0], want.astype(np.int64))
    a.close(); b.close()


FAKE_RCCL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "libfake_rccl.so")


def _native_worker(rank, world, port, first_tokens, L, D, seed, steps, prompt, q, rccl_lib):
    try:
        if rccl_lib:
            os.environ["RWKV_RCCL_LIB"] = rccl_lib          # engine.hip pipe_open: dlopen()s this instead of librccl.so
        import torch
        import torch.distributed as dist
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from rwkv_cpp_accelerated_amd import pipeline
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        l0, l1 = pipeline.partition_layers(L, world, D)[rank]
        st = pipeline.EngineStage(mf.synthetic_tensors(L, D, seed=seed), L, D, l0, l1, n_slots=world, device=0, prefill=True)
        pipeline.pipe_connect(st, dist, rank, world)
        info = st.m.pipe_info()
        pipeline.run_prefill_native(st, rank, prompt, len(prompt))
        lg = st.m.logits(32)[: mf.VOCAB * ((len(prompt) - 1) % 32 + 1)].reshape(-1, mf.VOCAB)[-1].copy() if rank == world - 1 else None
        picks = pipeline.run_pipeline_native(st, rank, world, first_tokens, steps)
        # ONE stream through all the stages (bench.py's `one_stream` leg), with the hop timing on: fresh state, stream 0 only
        st.m.reset_state()
        st.m.pipe_profile(True)
        picks1 = pipeline.run_pipeline_native(st, rank, world, first_tokens, steps, n_streams=1)
        hop = st.m.pipe_hop_stats()
        st.m.pipe_profile(False)
        bad = None
        if rank == 0:                                    # a bad id must fail on rank 0 WITHOUT stranding the other ranks
            try:
                pipeline.run_prefill_native(st, rank, [5, mf.VOCAB + 7, 9] + prompt[:40], 43)
            except Exception as e:                       # noqa: BLE001
                bad = str(e)
        else:
            pipeline.run_prefill_native(st, rank, None, 43)
        # 2 x world streams on two communicators (rwkv_pipe_decode_dual): fresh state, every rank takes part
        st.m.reset_state()
        first2 = [(7 * g + 3) % 50000 + 2 for g in range(2 * world)]
        picks2 = pipeline.run_pipeline_native_dual(st, rank, world, first2, steps)
        # ... and 20 more times on the same two communicators (a full-suite run once saw this leg time out): every repetition
        # must finish and give the same picks -- a lost wake-up or an ordering hole between the comm streams and the compute stream shows here
        for rep in range(20):
            st.m.reset_state()
            again = pipeline.run_pipeline_native_dual(st, rank, world, first2, steps)
            if rank == world - 1 and not np.array_equal(again, picks2):
                raise RuntimeError(f"two-communicator schedule, repetition {rep}: picks differ from the first run")
        if rank == 0:
            q.put(("rank0", bad))
        if rank == world - 1:
            q.put(("ok", picks, lg, picks1, hop, info, picks2, first2))
        dist.barrier()
        st.m.close()
        dist.destroy_process_group()
    except Exception as e:                         # pragma: no cover
        q.put(("error", rank, repr(e)))


@pytest.mark.parametrize("world", [2, 4])
def test_native_transport_with_several_ranks_on_one_gpu(eng_mod, world):
    """rwkv_pipe_init / rwkv_pipe_prefill / rwkv_pipe_decode EXECUTED with world = 2 and 4: one process per rank, all on cuda:0,
    the engine's ncclSend / ncclRecv groups served by tests/fake_rccl.cpp (shared-memory channels, stream ordered; real RCCL
    refuses two ranks on one device and a gpurun box has one GPU).  Exercises what world = 1 cannot: the composition of the
    per-tick group, the (ci - 1) & 1 / ci & 1 buffer parity of the prefill hop, the recv of the fed-back id into the control
    block behind its memcpy, the item order of the picks.  Picks and last-chunk logits must be bit-identical to a single
    whole-model context; a bad token id fails on rank 0 without stranding the other ranks.  Then the two-communicator schedule
    (rwkv_pipe_decode_dual: 2 x world streams, one parity's hop under the other's stage): every stream's picks equal a decode alone."""
    import torch.multiprocessing as mp
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/_build/libfake_rccl.so is missing: run __graft_entry__.build()")
    L, D, seed, steps = 8, 768, 93, 5
    first = [11, 222, 3333, 44444][:world]
    prompt = [int(x) for x in np.random.default_rng(5).integers(2, mf.VOCAB, 150)]        # 5 chunks: more chunks than stages, ragged tail
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, first, L, D, seed, steps, prompt, q, FAKE_RCCL)) for r in range(world)]
    [p.start() for p in procs]
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=420)
            res[r[0]] = r
            if r[0] == "error":
                break
    finally:
        [p.join(timeout=90) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
    assert "error" not in res, res["error"]
    assert "out of range" in (res["rank0"][1] or ""), res["rank0"]
    _, picks, lg, picks1, hop, info, picks2, first2 = res["ok"]
    # rwkv_pipe_info: what the first run on real xGMI will be diagnosed from (here: the stand-in's version code 1 and its path)
    assert info["rank"] == world - 1 and info["world"] == world and info["prefill_rows"] == 64 and info["rccl_version"] == 1, info
    assert info["rccl_path"].endswith("libfake_rccl.so") and info["device"] == 0 and info["arch"].startswith("gfx"), info
    t = mf.synthetic_tensors(L, D, seed=seed)
    m = eng_mod.RWKV(resident=True); m.loadTensors(L, D, t, maxGPT=32)
    for i in range(0, len(prompt), 32):
        ref = m.forward(prompt[i:i + 32], eng_mod.MODE_GPT)[: len(prompt[i:i + 32]) * mf.VOCAB].reshape(-1, mf.VOCAB)[-1].copy()
    assert np.array_equal(lg, ref)
    # the streams of the native decode started from slot k's state: slot 0 holds the prompt, the others are fresh
    for k, tk in enumerate(first):
        if k >= 1:
            m.reset_state()
        cur, ids = tk, []
        for _ in range(steps):
            cur = parity.argmax_ban0(m.forward(cur)[: mf.VOCAB]); ids.append(cur)
        assert list(picks[k]) == ids, k
    # rwkv_pipe_decode_streams with n_streams = 1: stream 0 from a fresh state, the other streams' rows untouched, and the last rank's
    # receives (x from rank world - 2, every step) bracketed by event pairs
    m.reset_state()
    cur, ids = first[0], []
    for _ in range(steps):
        cur = parity.argmax_ban0(m.forward(cur)[: mf.VOCAB]); ids.append(cur)
    assert list(picks1[0]) == ids and not picks1[1:].any(), picks1
    assert hop["n"] == steps and 0.0 < hop["min_us"] <= hop["mean_us"] <= hop["max_us"], hop
    # rwkv_pipe_decode_dual: 2 x world independent streams, two per stage in flight on two communicators; every stream from a fresh state
    assert picks2.shape == (2 * world, steps)
    for g, tk in enumerate(first2):
        m.reset_state()
        cur, ids = tk, []
        for _ in range(steps):
            cur = parity.argmax_ban0(m.forward(cur)[: mf.VOCAB]); ids.append(cur)
        assert list(picks2[g]) == ids, (g, list(picks2[g]), ids)
    m.close()



def _mismatch_worker(rank, world, port, L, D, q, rccl_lib):
    try:
        os.environ["RWKV_RCCL_LIB"] = rccl_lib
        if rank == 1:
            os.environ["RWKV_SEQ_ROWS"] = "32"           # a per-rank environment difference: this rank would cut prompts into 32-row micro-batches
        import torch.distributed as dist
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from rwkv_cpp_accelerated_amd import pipeline
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        l0, l1 = pipeline.partition_layers(L, world, D)[rank]
        st = pipeline.EngineStage(mf.synthetic_tensors(L, D, seed=7), L, D, l0, l1, n_slots=world, device=0, prefill=True)
        err = None
        try:
            pipeline.pipe_connect(st, dist, rank, world)
        except Exception as e:                           # noqa: BLE001
            err = str(e)
        q.put(("rank", rank, err))
        dist.barrier()
        st.m.close()
        dist.destroy_process_group()
    except Exception as e:                               # pragma: no cover
        q.put(("error", rank, repr(e)))


def test_pipe_init_refuses_ranks_that_disagree_on_the_micro_batch(eng_mod):
    """rwkv_pipe_prefill's micro-batch (64 or 32 rows) comes from per-rank values (RWKV_SEQ_ROWS, max_ctx): ranks that differ would
    exchange differently sized messages and hang or corrupt the residual stream.  rwkv_pipe_init agrees on it over the communicator
    (ADVICE r04): with RWKV_SEQ_ROWS=32 on rank 1 only, EVERY rank must fail the init with the same diagnosis."""
    import torch.multiprocessing as mp
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/_build/libfake_rccl.so is missing: run __graft_entry__.build()")
    world, L, D = 2, 4, 768
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, world, port, L, D, q, FAKE_RCCL)) for r in range(world)]
    [p.start() for p in procs]
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=300)
            assert r[0] != "error", r
            got[r[1]] = r[2]
    finally:
        [p.join(timeout=60) for p in procs]
        for p in procs:
            if p.is_alive():
                p.kill()
    for r in range(world):
        assert got[r] and "disagree" in got[r] and "32..64" in got[r], got
