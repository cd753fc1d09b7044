"""CPU checks of the oracle (the C restatement of rwkv.cu:493-593) against independent numpy
float64 math, of the format helpers, and of the C-ABI surface.  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mm8_one_matches_float64(oracle):
    rng = np.random.default_rng(0)
    for N, M in [(64, 48), (768, 768), (256, 1024), (1024, 256), (128, 50277 // 16)]:
        x = rng.standard_normal((2, N))
        w = rng.integers(0, 256, (N, M), dtype=np.uint8)
        a = (1.0 / np.sqrt(N)) * (0.5 + rng.random(N))
        r = (2 * a / 255).astype(np.float32); o = (-a).astype(np.float32)
        y = oracle.mm8_one(x, w, r, o)
        ref = x.astype(np.float32).astype(np.float64) @ (w.astype(np.float64) * r[:, None].astype(np.float64) + o[:, None].astype(np.float64))
        scale = np.abs(ref).max()
        assert np.abs(y - ref).max() <= 2e-5 * scale
        # f32 input variant (ffn_v path, rwkv.cu:574) and accumulate-onto-y semantics (rwkv.cu:548-550)
        y2 = oracle.mm8_one(x.astype(np.float32), w, r, o, y0=np.ones((2, M), np.float32))
        assert np.abs(y2 - 1.0 - ref).max() <= 2e-5 * scale + 1e-6


def test_layernorm_semantics(oracle):
    """unbiased variance (D-1), no epsilon (rwkv.cu:43-44,53)"""
    rng = np.random.default_rng(1)
    D = 768
    x = rng.standard_normal((3, D)) * 3 + 0.5
    ln = np.stack([1 + 0.1 * rng.standard_normal(D), 0.1 * rng.standard_normal(D)])
    out = oracle.layernorm(x, ln)
    ref = ln[0] * (x - x.mean(1, keepdims=True)) / x.std(1, ddof=1, keepdims=True) + ln[1]
    assert np.abs(out - ref).max() < 5e-6
    # and it is NOT the torch convention (biased variance + eps) at this tolerance
    torch_like = ln[0] * (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) + ln[1]
    assert np.abs(out - torch_like).max() > 1e-4


def test_wkv_recurrence(oracle):
    """un-stabilised f64 recurrence with post-decay state (rwkv.cu:242-255)"""
    rng = np.random.default_rng(2)
    Cn, T = 96, 5
    w = -np.exp(rng.uniform(-6, 1, Cn)); u = 0.3 * rng.standard_normal(Cn)
    k = rng.standard_normal((T, Cn)).astype(np.float32); v = rng.standard_normal((T, Cn)).astype(np.float32)
    r = rng.standard_normal((T, Cn)).astype(np.float32)
    aa = np.zeros(Cn); bb = np.zeros(Cn); pp = np.full(Cn, -1e30)
    y = oracle.wkv(w, u, k, v, r, aa, bb, pp)
    A = np.zeros(Cn); B = np.zeros(Cn)
    for t in range(T):
        kk, vv, rr = k[t].astype(np.float64), v[t].astype(np.float64), r[t].astype(np.float64)
        e = np.exp(u + w + kk)
        yy = (A + e * vv) / (B + e) / (1 + np.exp(-rr))
        assert np.abs(y[t] - yy).max() < 1e-6 * max(1.0, np.abs(yy).max())
        A = (A + np.exp(kk) * vv) * np.exp(w); B = (B + np.exp(kk)) * np.exp(w)
    assert np.allclose(aa, A, rtol=1e-12) and np.allclose(bb, B, rtol=1e-12)
    assert np.all(pp == -1e30)   # pp is carried through unchanged (rwkv.cu:244,255)


def test_quantizer_roundtrip(oracle):
    """quantize_matrix (convert_model.py:108-119): per-input-column asymmetric u8"""
    rng = np.random.default_rng(3)
    W = rng.standard_normal((40, 24)).astype(np.float32)   # torch layout [out][in]
    q, r, o = oracle.quantize_matrix(W)
    assert q.shape == (24, 40) and q.dtype == np.uint8
    deq = q.astype(np.float64) * r[:, None] + o[:, None]
    assert np.abs(deq - W.T).max() <= 1.01 * r.max()
    # truncation-bias compensation: mean dequantisation error per input row ~ 0
    assert np.abs((deq - W.T).mean(1)).max() < 1e-6 + 0.02 * r.max()


def test_modelfile_layout(tmp_path, oracle):
    L, D = 2, 64
    assert mf.file_bytes(L, D) == 16429028            # size verified against the reference converter
    t = mf.synthetic_tensors(L, D, seed=5)
    p = str(tmp_path / "m.bin")
    mf.write_bin(p, L, D, t)
    assert os.path.getsize(p) == mf.file_bytes(L, D)
    a, b, back = mf.read_bin(p)
    assert (a, b) == (L, D)
    for i in range(mf.N_TENSORS):
        assert back[i].dtype == mf.DTYPES[i] and np.array_equal(np.asarray(back[i]), np.asarray(t[i]).reshape(-1))
        assert oracle.L.oracle_tensor_elems(i, L, D) == mf.sizes(L, D)[i]
        assert oracle.L.oracle_tensor_type(i) == np.dtype(mf.DTYPES[i]).itemsize
    # the oracle's own reader sees the same model as the in-memory tensors
    m1 = oracle.open_file(p); m2 = oracle.from_tensors(L, D, t)
    toks = [3, 17, 50276]
    l1 = m1.forward(toks, m1.new_state()); l2 = m2.forward(toks, m2.new_state())
    assert np.array_equal(l1, l2)
    m1.close(); m2.close()


def test_oracle_modes_and_chunking(oracle):
    """GPT mode over T tokens == T single-token calls; PARRALEL mode == independent sequences
    (rwkv.cu:236-240 state indexing)."""
    L, D = 2, 64
    t = mf.synthetic_tensors(L, D, seed=7)
    m = oracle.from_tensors(L, D, t)
    toks = [5, 9, 1234, 77]
    s_a = m.new_state(); la = m.forward(toks, s_a)
    s_b = m.new_state(); lb = np.concatenate([m.forward([tk], s_b) for tk in toks])
    assert np.array_equal(la, lb) and all(np.array_equal(x, y) for x, y in zip(s_a, s_b))
    s_p = m.new_state(slots=len(toks)); lp = m.forward(toks, s_p, mode=0)
    for i, tk in enumerate(toks):
        s1 = m.new_state(); l1 = m.forward([tk], s1)
        assert np.array_equal(lp[i], l1[0])
        for arr_p, arr_1 in zip(s_p, s1):
            assert np.array_equal(arr_p[i * L * D:(i + 1) * L * D], arr_1)
    m.close()


def test_abi_exports_every_declared_symbol(built):
    """the C-ABI library loads on a box without a GPU and exports every symbol include/*.h declares"""
    hdr = open(os.path.join(ROOT, "include", "rwkv_mi355x.h")).read()
    declared = sorted(set(re.findall(r"\b(rwkv_[a-z0-9_]+)\s*\(", hdr)))
    lib = C.CDLL(built["engine"])
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    from rwkv_cpp_accelerated_amd import engine
    assert sorted(engine.ABI_SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu(built):
    """the product path fails loudly when there is no HIP device (never routes through the oracle)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rwkv_cpp_accelerated_amd import engine
    with pytest.raises(engine.RWKVError, match="no HIP device"):
        engine.RWKV()
