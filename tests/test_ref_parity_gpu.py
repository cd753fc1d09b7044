"""Pin the oracle AND the engine against the reference's own kernel: include/rwkv/cuda/rwkv.cu built
unmodified with hipcc (oracle/_ref/libref.so, built in the authoring container, shipped to the GPU
box).  The reference is run-to-run nondeterministic at ~1e-6 (float atomicAdd order, rwkv.cu:95,292)."""
import os

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf
import oracle_lib
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(oracle_lib.REF_SO):
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference at build time)")
    return oracle_lib.Ref()


@pytest.mark.parametrize("L,D", [(2, 768), (2, 2048)])
def test_oracle_and_engine_vs_reference_kernel(built, oracle, ref, tmp_path, L, D):
    from rwkv_cpp_accelerated_amd import engine
    t = mf.synthetic_tensors(L, D, seed=21 + D)
    p = str(tmp_path / "model.bin")
    mf.write_bin(p, L, D, t)
    rm = ref.load_file(p, 1)
    om = oracle.open_file(p)
    em = engine.RWKV(resident=True); em.loadFile(p)
    st = om.new_state()
    tk = 17
    for step in range(12):
        lr = rm.forward([tk])[0]
        lo = om.forward([tk], st)[0]
        le = em.forward(tk)[: mf.VOCAB]
        parity.check_logits(lo, lr, f"oracle vs ref, step {step}")
        parity.check_logits(le, lr, f"engine vs ref, step {step}")
        parity.check_argmax(le, lr, f"engine vs ref, step {step}")
        tk = parity.argmax_ban0(lr)           # teacher-forced on the reference's greedy ids
    for i, s in enumerate(st):               # oracle state vs the reference's host-authoritative state
        r = rm.state(i)
        assert np.abs(s - r).max() <= 1e-4 * max(1.0, np.abs(r).max())
    om.close(); em.close()


def test_reference_gpt_chunk(built, oracle, ref, tmp_path):
    """multi-token GPT-mode call of the reference (rwkv.h:395-413 loadContext path) vs oracle and engine"""
    from rwkv_cpp_accelerated_amd import engine
    L, D, T = 2, 768, 4
    t = mf.synthetic_tensors(L, D, seed=33)
    p = str(tmp_path / "model.bin")
    mf.write_bin(p, L, D, t)
    rm = ref.load_file(p, T); om = oracle.open_file(p)
    em = engine.RWKV(resident=True); em.loadFile(p, T)
    toks = [100, 200, 300, 400]
    lr = rm.forward(toks); lo = om.forward(toks, om.new_state())
    le = em.forward(toks, engine.MODE_GPT)[: T * mf.VOCAB].reshape(T, mf.VOCAB)
    for i in range(T):
        parity.check_logits(lo[i], lr[i], f"oracle pos {i}")
        parity.check_logits(le[i], lr[i], f"engine pos {i}")
    om.close(); em.close()


@pytest.mark.parametrize("name,steps", [("1B5", 96), ("7B", 64), ("14B", 64)])
def test_full_depth_gate_vs_reference_kernel(built, ref, name, steps):
    """BASELINE configs 2, 3 and 4 at FULL depth (1B5: L=24, D=2048; 7B: L=32, D=4096; 14B: L=40, D=5120): the engine,
    teacher-forced with the reference kernel's greedy ids on the same device-resident tensors, matches its logits within 1e-3 and
    its greedy id at every step (the 1024-step instance of the 7B gate runs inside bench.py: parity_vs_reference_kernel, and
    bench.py exits non-zero when it trips)."""
    import torch
    from rwkv_cpp_accelerated_amd import engine
    import refgate
    L, D = mf.SHAPES[name]
    t = mf.synthetic_tensors_torch(L, D, seed=5, device="cuda")
    torch.cuda.synchronize()
    em = engine.RWKV(resident=True); em.loadTensors(L, D, t, maxGPT=1)
    rm = refgate.ref_model_from_torch(ref, mf, t, L, D, 1)
    rng = np.random.default_rng(3)
    prompt = [int(x) for x in rng.integers(2, mf.VOCAB, 8)]
    g = refgate.run_gate(rm, em, mf, prompt, steps, strict=True, what=name)
    assert g["steps"] == steps and g["steps_outside_tolerance"] == 0
    assert len(set(g["ids"])) > 4, "degenerate greedy chain: the gate would not exercise the recurrence"
    em.close()
    del t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,T", [("14B", 32), ("7B", 32), ("1B5", 32)])
def test_chunk_path_full_depth_vs_reference_kernel(built, ref, name, T):
    """BASELINE config 5 (7B, T = 32) and its shape at 14B / 1B5, FULL depth (also inside bench.py: prefill.parity_vs_reference_kernel):
    a T-token prompt as ONE GPT-mode call of the reference's own kernel (rwkv.h:339-376; in-kernel token loops rwkv.cu:227,279)
    against the engine's chunk path -- all T logits rows, the five state arrays, 8 greedy decode steps from that state -- and two
    T-slot PARRALEL steps (rwkv.cu:236-240), all rows, all slots of the state."""
    import torch
    from rwkv_cpp_accelerated_amd import engine
    import refgate
    L, D = mf.SHAPES[name]
    t = mf.synthetic_tensors_torch(L, D, seed=6, device="cuda")
    torch.cuda.synchronize()
    em = engine.RWKV(resident=True); em.loadTensors(L, D, t, maxGPT=T)
    rm = refgate.ref_model_from_torch(ref, mf, t, L, D, T)
    prompt = [int(x) for x in np.random.default_rng(8).integers(2, mf.VOCAB, T)]
    g = refgate.run_chunk_gate(rm, em, mf, engine, prompt, decode_steps=8, strict=True, what=name)
    assert g["gpt_chunk"]["rows_outside_tolerance"] == 0 and g["gpt_chunk"]["decode_steps_outside_tolerance"] == 0
    assert g["parralel_step"]["rows_outside_tolerance"] == 0
    em.close()
    del t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,T", [("1B5", 160), ("3B", 96)])
def test_long_prompt_vs_reference_kernel(built, ref, name, T):
    """A prompt of SEVERAL weight passes in ONE rwkv_forward call -- 64-row passes (the last one ragged at T = 160: 64 + 64 + 32;
    64 + 32 at T = 96), the three-stream software pipeline, the captured pass graphs: the schedule bench.py's `long_prompt` leg
    times -- at FULL depth against the reference's own kernel fed the same tokens in GPT-mode calls of 32 (rwkv.h:395-413,
    rwkv.cu:227,279): every logits row, the five state arrays, 4 greedy decode steps from that state."""
    import torch
    from rwkv_cpp_accelerated_amd import engine
    import refgate
    L, D = mf.SHAPES[name]
    t = mf.synthetic_tensors_torch(L, D, seed=9, device="cuda")
    torch.cuda.synchronize()
    em = engine.RWKV(resident=True); em.loadTensors(L, D, t, maxGPT=T)
    rm = refgate.ref_model_from_torch(ref, mf, t, L, D, 32)
    toks = [int(x) for x in np.random.default_rng(10).integers(2, mf.VOCAB, T)]
    g = refgate.run_long_prompt_gate(rm, em, mf, engine, toks, ref_chunk=32, decode_steps=4, strict=True, what=name)
    assert g["rows"] == T and g["rows_outside_tolerance"] == 0 and g["decode_steps_outside_tolerance"] == 0
    em.close()
    del t
    torch.cuda.empty_cache()


def test_96_streams_vs_reference_kernel(built, ref):
    """96 PARRALEL-mode streams per step = a 64-row and a 32-row weight pass pipelined over the engine's streams (bench.py's
    `batched_decode.streams_96` leg) at FULL depth (1B5) against the reference kernel's 96-slot step (rwkv.cu:236-240), two rounds:
    every logits row and all 96 slots of the five state arrays."""
    import torch
    from rwkv_cpp_accelerated_amd import engine
    import refgate
    L, D = mf.SHAPES["1B5"]
    t = mf.synthetic_tensors_torch(L, D, seed=12, device="cuda")
    torch.cuda.synchronize()
    em = engine.RWKV(resident=True); em.loadTensors(L, D, t, maxGPT=96)
    rm = refgate.ref_model_from_torch(ref, mf, t, L, D, 96)
    first = [int(x) for x in np.random.default_rng(13).integers(2, mf.VOCAB, 96)]
    g = refgate.run_streams_gate(rm, em, mf, engine, first, rounds=2, strict=True, what="1B5 x 96")
    assert g["slots"] == 96 and g["rows_outside_tolerance"] == 0
    em.close()
    del t
    torch.cuda.empty_cache()
