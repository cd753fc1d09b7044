"""GPU parity tests of the chunked prompt path (BASELINE config 5: batch-32 prefill, mm8_seq on the
int8 matrix cores; csrc/seq.hip.h) against the CPU oracle, which runs the reference's GPT-mode
semantics token by token (rwkv.cu:493-593 with tokenlength > 1)."""
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


def _toks(n, seed):
    return [int(v) for v in np.random.default_rng(seed).integers(2, mf.VOCAB, n)]


# ragged chunk, exactly one MFMA tile, exactly 32, more than one chunk, the widest / narrowest models
@pytest.mark.parametrize("L,D,T", [(2, 768, 5), (2, 768, 16), (2, 1024, 32), (1, 2048, 33), (1, 2560, 2), (1, 4096, 32), (1, 5120, 17), (2, 64, 40)])
def test_chunk_logits_and_state_vs_oracle(eng_mod, oracle, L, D, T):
    t = mf.synthetic_tensors(L, D, seed=300 + D + T)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t, maxGPT=max(T, 2))
    om = oracle.from_tensors(L, D, t)
    st = om.new_state()
    toks = _toks(T, D + T)
    ref = om.forward(toks, st)
    got = m.forward(toks, eng_mod.MODE_GPT)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
    for i in range(T):
        parity.check_logits(got[i], ref[i], f"L{L} D{D} pos {i}")
        parity.check_argmax(got[i], ref[i], f"L{L} D{D} pos {i}")
    m.pull_state(1)
    for name, g, r in zip("xy aa bb pp dd".split(), m.state.arrays(), st):
        assert np.abs(g[: L * D] - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), name
    # decoding continues from the chunk's state (loadContext then generate, rwkv.h:395-413)
    for step, tk in enumerate([7, 4242, 50276]):
        refs = om.forward([tk], st)[0]
        gots = m.forward(tk)[: mf.VOCAB]
        parity.check_logits(gots, refs, f"L{L} D{D} continuation {step}")
    om.close(); m.close()


def test_chunk_is_deterministic(eng_mod):
    L, D, T = 2, 1024, 32
    t = mf.synthetic_tensors(L, D, seed=77)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t, maxGPT=T)
    toks = _toks(T, 5)
    a = m.forward(toks, eng_mod.MODE_GPT)[: T * mf.VOCAB].copy()
    m.reset_state()
    b = m.forward(toks, eng_mod.MODE_GPT)[: T * mf.VOCAB].copy()
    assert np.array_equal(a, b)      # integer contraction + exact LDS sums: no order dependence
    m.close()


@pytest.mark.parametrize("L,D,T", [(2, 768, 3), (2, 1024, 32), (1, 4096, 17), (2, 256, 40)])
def test_parralel_batch_vs_oracle(eng_mod, oracle, L, D, T):
    """PARRALEL mode (rwkv.cu:236-240): T independent sequences advance one token per call, state slot t;
    the engine runs the batch through the MFMA path (weights read once for all streams)."""
    t = mf.synthetic_tensors(L, D, seed=500 + D + T)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t, maxGPT=T)
    om = oracle.from_tensors(L, D, t)
    sp = om.new_state(slots=T)
    for rnd in range(3):                     # the state of every slot carries over between rounds
        toks = _toks(T, 1000 * rnd + T)
        ref = om.forward(toks, sp, mode=0)
        got = m.forward(toks, eng_mod.MODE_PARRALEL)[: T * mf.VOCAB].reshape(T, mf.VOCAB)
        for i in range(T):
            parity.check_logits(got[i], ref[i], f"L{L} D{D} round {rnd} slot {i}")
            parity.check_argmax(got[i], ref[i], f"L{L} D{D} round {rnd} slot {i}")
    m.pull_state(T)
    for name, g, r in zip("xy aa bb pp dd".split(), m.state.arrays(), sp):
        assert np.abs(g[: T * L * D] - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), name
    om.close(); m.close()
