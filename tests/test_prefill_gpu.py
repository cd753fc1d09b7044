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


@pytest.mark.parametrize("L,D,T", [(2, 768, 3), (2, 1024, 32), (1, 4096, 17), (2, 256, 40), (2, 768, 70), (1, 1024, 96)])
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


def _adversarial(L, D, seed):
    """synthetic model with the statistics that stress the fixed-point paths: embedding rows with a mean far
    from zero, LayerNorm weights / biases with outlier channels, token-shift mixes pinned at 0 and 1, scale
    vectors spanning two decades, extreme decays"""
    t = mf.synthetic_tensors(L, D, seed=seed)
    rng = np.random.default_rng(seed + 1)
    t[mf.EMBED] = (t[mf.EMBED].reshape(mf.VOCAB, D) * 5.0 + 20.0).astype(np.float32).reshape(-1)
    ln = t[mf.LAYERNORMS].reshape(-1, D).copy()
    hot = rng.random(ln.shape) < 0.01
    ln[0::2][hot[0::2]] *= 20.0
    ln[1::2][hot[1::2]] += 10.0 * rng.choice([-1.0, 1.0], size=int(hot[1::2].sum()))
    t[mf.LAYERNORMS] = ln.reshape(-1)
    for s in (mf.MIXK, mf.MIXV, mf.MIXR, mf.FFNMIXK, mf.FFNMIXV):
        v = t[s].copy(); u = rng.random(v.shape)
        v[u < 0.05] = 0.0; v[u > 0.95] = 1.0
        t[s] = v
    for s in (mf.KR, mf.VR, mf.RR, mf.ATTOUTR, mf.FFNKR, mf.FFNVR, mf.FFNRR):
        t[s] = (t[s] * np.exp(rng.uniform(np.log(0.3), np.log(3.0), t[s].shape))).astype(np.float32)
    d = t[mf.DECAY].copy(); u = rng.random(d.shape)
    d[u < 0.02] = -20.0; d[u > 0.98] = -1e-4
    t[mf.DECAY] = d
    return t


@pytest.mark.parametrize("L,D", [(2, 1024), (1, 4096)])
def test_adversarial_statistics_decode_and_chunk(eng_mod, oracle, L, D):
    t = _adversarial(L, D, seed=900 + D)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t, maxGPT=16)
    om = oracle.from_tensors(L, D, t)
    st = om.new_state()
    for step, tk in enumerate(_toks(6, D)):                 # single-token kernels (bound-based scales, site tables)
        ref = om.forward([tk], st)[0]
        parity.check_logits(m.forward(tk)[: mf.VOCAB], ref, f"adversarial D{D} decode {step}")
    toks = _toks(16, D + 1)                                 # then a chunk from that state (per-token exact scales)
    ref = om.forward(toks, st)
    got = m.forward(toks, eng_mod.MODE_GPT)[: 16 * mf.VOCAB].reshape(16, mf.VOCAB)
    for i in range(16):
        parity.check_logits(got[i], ref[i], f"adversarial D{D} chunk pos {i}")
    m.pull_state(1)
    for name, g, r in zip("xy aa bb pp dd".split(), m.state.arrays(), st):
        assert np.abs(g[: L * D] - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), name
    om.close(); m.close()


@pytest.mark.parametrize("L,D,T,mode", [(4, 768, 200, "gpt"), (3, 1024, 97, "gpt"), (2, 768, 70, "par")])
def test_long_prompt_two_stage_pipeline_is_bit_identical(eng_mod, oracle, L, D, T, mode, monkeypatch):
    """A forward call of several chunks runs as a two-stage software pipeline on two streams (engine.hip rwkv_forward: layers
    [0, mid) of chunk i + 1 under layers [mid, L) + head of chunk i).  Same kernels on the same data: every logits row and the
    whole recurrent state must equal the one-stream schedule (RWKV_SEQ_STAGES=1) bit for bit -- and the oracle within tolerance."""
    t = mf.synthetic_tensors(L, D, seed=700 + D + T)
    toks = _toks(T, 5 * D + T)
    md = eng_mod.MODE_GPT if mode == "gpt" else eng_mod.MODE_PARRALEL
    outs = {}
    for split in ("0", "1"):
        monkeypatch.setenv("RWKV_SEQ_STAGES", "1" if split == "0" else "3")
        m = eng_mod.RWKV(resident=True)
        m.loadTensors(L, D, t, maxGPT=T)
        lg = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        lg2 = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()      # a second call: the pipeline's buffers and events are reused
        m.pull_state(T if mode == "par" else 1)
        outs[split] = (lg, lg2, [a.copy() for a in m.state.arrays()])
        m.close()
    assert np.array_equal(outs["0"][0], outs["1"][0])
    assert np.array_equal(outs["0"][1], outs["1"][1])
    for a, b in zip(outs["0"][2], outs["1"][2]):
        assert np.array_equal(a, b)
    if mode == "gpt":
        om = oracle.from_tensors(L, D, t)
        st = om.new_state()
        ref = om.forward(toks, st)
        for i in (0, 31, 32, 63, 64, T - 1):
            parity.check_logits(outs["1"][0][i], ref[i], f"pos {i}")
        om.close()


@pytest.mark.parametrize("mode,T", [("gpt", 150), ("par", 96)])
def test_seq_stages_1_to_4_are_bit_identical_and_match_the_oracle(eng_mod, oracle, mode, T, monkeypatch):
    """RWKV_SEQ_STAGES = 1, 2, 3 (default), 4: the software pipeline over the chunks of one forward call only re-schedules the same
    kernels on the same data -- every logits row and the whole recurrent state are bit-identical across the four settings; the
    result is the ORACLE's within tolerance in both modes (PARRALEL with more than 32 slots = several passes)."""
    L, D = 4, 768
    t = mf.synthetic_tensors(L, D, seed=811 + T)
    toks = _toks(T, 3 * T)
    md = eng_mod.MODE_GPT if mode == "gpt" else eng_mod.MODE_PARRALEL
    outs = {}
    for stages in ("1", "2", "3", "4"):
        monkeypatch.setenv("RWKV_SEQ_STAGES", stages)
        m = eng_mod.RWKV(resident=True)
        m.loadTensors(L, D, t, maxGPT=T)
        lg = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        lg2 = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()      # second call: buffers and events reused
        m.pull_state(T if mode == "par" else 1)
        outs[stages] = (lg, lg2, [a.copy() for a in m.state.arrays()])
        m.close()
    for stages in ("2", "3", "4"):
        assert np.array_equal(outs["1"][0], outs[stages][0]), stages
        assert np.array_equal(outs["1"][1], outs[stages][1]), stages
        for a, b in zip(outs["1"][2], outs[stages][2]):
            assert np.array_equal(a, b), stages
    om = oracle.from_tensors(L, D, t)
    st = om.new_state(slots=T if mode == "par" else 1)
    ref = om.forward(toks, st, mode=0 if mode == "par" else 1)
    for i in range(T):
        parity.check_logits(outs["3"][0][i], ref[i], f"{mode} row {i}")
    ref2 = om.forward(toks, st, mode=0 if mode == "par" else 1)                       # the second call continues from the first one's state
    for i in (0, 31, 32, T - 1):
        parity.check_logits(outs["3"][1][i], ref2[i], f"{mode} second call row {i}")
    n = (T if mode == "par" else 1) * L * D
    for name, g, r in zip("xy aa bb pp dd".split(), outs["3"][2], st):
        assert np.abs(g[:n] - r[:n]).max() <= 1e-4 * max(1.0, np.abs(r).max()), name
    om.close()



@pytest.mark.parametrize("mode,T,L,D", [("gpt", 150, 3, 768), ("gpt", 64, 2, 2048), ("gpt", 47, 2, 1024), ("par", 96, 3, 768), ("par", 70, 2, 2560), ("gpt", 70, 1, 5120), ("gpt", 96, 1, 4096)])
def test_64_row_passes_are_bit_identical_to_32_row_chunks(eng_mod, oracle, mode, T, L, D, monkeypatch):
    """Round 4: a forward call of more than 32 rows runs in passes of up to 64 rows = TWO halves that share every weight fragment
    (seq.hip.h SEQ_TM; k_seq_gemm_p with NH = 2, the element-wise kernels on a global row index, k_seq_wkv over 64 rows) -- weights
    read once per 64 rows.  It is a re-scheduling of the same arithmetic: every logits row and the whole recurrent state must be
    bit-identical to the 32-row schedule (RWKV_SEQ_ROWS=32), for full and ragged passes (150 = 64 + 64 + 22, 47 = 32 + 15), in both
    modes, at 1 KiB-multiple and other row sizes; and they must be the ORACLE's within tolerance."""
    t = mf.synthetic_tensors(L, D, seed=640 + T)
    toks = _toks(T, 5 * T)
    md = eng_mod.MODE_GPT if mode == "gpt" else eng_mod.MODE_PARRALEL
    outs = {}
    for rows in ("32", "64"):
        monkeypatch.setenv("RWKV_SEQ_ROWS", rows)
        m = eng_mod.RWKV(resident=True)
        m.loadTensors(L, D, t, maxGPT=T)
        lg = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        lg2 = m.forward(toks, md)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        m.pull_state(T if mode == "par" else 1)
        outs[rows] = (lg, lg2, [a.copy() for a in m.state.arrays()])
        m.close()
    assert np.array_equal(outs["32"][0], outs["64"][0])
    assert np.array_equal(outs["32"][1], outs["64"][1])
    for a, b in zip(outs["32"][2], outs["64"][2]):
        assert np.array_equal(a, b)
    om = oracle.from_tensors(L, D, t)
    st = om.new_state(slots=T if mode == "par" else 1)
    ref = om.forward(toks, st, mode=0 if mode == "par" else 1)
    for i in sorted({0, 31, 32, 33, 46, 63, T - 1} & set(range(T))):
        parity.check_logits(outs["64"][0][i], ref[i], f"{mode} row {i}")
    om.close()


@pytest.mark.parametrize("T,L,D", [(70, 2, 2560), (96, 1, 4096)])
def test_64_row_gemm_forms_agree(eng_mod, T, L, D, monkeypatch):
    """The 64-row pass has two forms of its K/V/R and ffn k/r GEMMs: k_seq_gemm_p<.., true, 2> (image re-staged per two k-blocks, plain
    workgroup ranges; RWKV_SEQ_B=0) and k_seq_gemm_b (one vector's image resident, workgroup ranges aligned to the vector groups, a wave's
    tiles in batches; RWKV_SEQ_B=5 forces both kinds, the default picks by width).  Same arithmetic per output: identical logits and state."""
    t = mf.synthetic_tensors(L, D, seed=77 + T)
    toks = _toks(T, 3 * T)
    outs = {}
    for b in ("0", "5", None):
        if b is None:
            monkeypatch.delenv("RWKV_SEQ_B", raising=False)
        else:
            monkeypatch.setenv("RWKV_SEQ_B", b)
        m = eng_mod.RWKV(resident=True)
        m.loadTensors(L, D, t, maxGPT=T)
        lg = m.forward(toks, eng_mod.MODE_GPT)[: T * mf.VOCAB].reshape(T, mf.VOCAB).copy()
        m.pull_state(1)
        outs[b] = (lg, [a.copy() for a in m.state.arrays()])
        m.close()
    for b in ("5", None):
        assert np.array_equal(outs["0"][0], outs[b][0]), b
        for x, y in zip(outs["0"][1], outs[b][1]):
            assert np.array_equal(x, y), b


@pytest.mark.parametrize("T,L,D", [(150, 3, 768), (64, 2, 2048), (20, 2, 1024)])
def test_captured_passes_replay_bit_identically(eng_mod, T, L, D, monkeypatch):
    """GPT-mode passes are captured as hipGraphs at first use (one per layer range, residual buffer, row count and logits row) and
    replayed afterwards (engine.hip enqueue_pass; RWKV_GRAPH bit 1 clear: direct launches).  First call (capture + launch), second call
    (replay) and the direct launches must agree bit for bit, logits and state, also when a different prompt goes through the same graphs."""
    t = mf.synthetic_tensors(L, D, seed=900 + T)
    toks, toks2 = _toks(T, 7 * T), _toks(T, 11 * T)
    outs = {}
    for gr in ("0", "1"):
        monkeypatch.setenv("RWKV_GRAPH", "1" if gr == "0" else "3")
        m = eng_mod.RWKV(resident=True)
        m.loadTensors(L, D, t, maxGPT=T)
        res = []
        for tk in (toks, toks, toks2, toks[: T // 2 + 1]):      # the last: a shorter call through the SAME context's graphs (other ragged pass)
            m.reset_state()
            lg = m.forward(tk, eng_mod.MODE_GPT)[: len(tk) * mf.VOCAB].reshape(len(tk), mf.VOCAB).copy()
            m.pull_state(1)
            res.append((lg, [a.copy() for a in m.state.arrays()]))
        outs[gr] = res
        m.close()
    for i in range(4):
        assert np.array_equal(outs["0"][i][0], outs["1"][i][0]), i
        for x, y in zip(outs["0"][i][1], outs["1"][i][1]):
            assert np.array_equal(x, y), i
    assert np.array_equal(outs["1"][0][0], outs["1"][1][0])
    assert not np.array_equal(outs["1"][0][0], outs["1"][2][0])
