"""The C++ drop-in header include/rwkv.h (class RWKV / RWKVState over the C-ABI) and the pybind module `rwkv`."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf
import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rwkv-cpp-accelerated_amd", "csrc")


@pytest.fixture(scope="module")
def app(built, tmp_path_factory):
    """compiles on a box without a GPU: proves the header is self-contained and links against the C-ABI"""
    exe = str(tmp_path_factory.mktemp("cpp") / "greedy_app")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "greedy_app.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + CSRC, "-lrwkv_mi355x", "-Wl,-rpath," + CSRC, "-o", exe])
    return exe


def test_cpp_header_compiles_and_links(app):
    assert os.path.exists(app)


def test_pybind_module_surface(built):
    """all 11 names of bindings/pybind/c_binding.cpp:158-175: the 8 model functions and the 3 tokenizer forwards"""
    assert built["pybind"], "pybind module not built"
    sys.path.insert(0, CSRC)
    import rwkv
    for f in ("initRwkv", "loadModel", "modelForward", "initState", "getState", "initOutput", "getOutput", "typicalSample"):
        assert hasattr(rwkv, f), f


@pytest.mark.gpu
def test_cpp_app_matches_python_engine(app, tmp_path):
    from rwkv_cpp_accelerated_amd import engine
    L, D = 2, 768
    t = mf.synthetic_tensors(L, D, seed=77)
    p = str(tmp_path / "model.bin")
    mf.write_bin(p, L, D, t)
    out = subprocess.run([app, p, "42", "12"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip() and not l.startswith(("n_layers", "n_embed"))]
    ids = [int(x) for x in lines[-2].split()]
    assert lines[-1].startswith("state_ok")
    m = engine.RWKV(resident=True); m.loadFile(p, 2)
    m.forward([5, 6], engine.MODE_GPT); m.forward([7], engine.MODE_GPT)
    tk, want = 42, []
    for _ in range(12):
        tk = parity.argmax_ban0(m.forward(tk)[: mf.VOCAB]); want.append(tk)
    assert ids == want
    m.close()


@pytest.mark.gpu
def test_pybind_module_forward(built, tmp_path):
    sys.path.insert(0, CSRC)
    import rwkv
    from rwkv_cpp_accelerated_amd import engine
    L, D = 2, 768
    t = mf.synthetic_tensors(L, D, seed=78)
    p = str(tmp_path / "model.bin")
    mf.write_bin(p, L, D, t)
    h = rwkv.initRwkv()
    assert rwkv.loadModel(h, p) == (L, D)
    rwkv.initOutput(h); rwkv.initState(h)
    m = engine.RWKV(resident=True); m.loadFile(p)
    for tk in (9, 8, 7):
        rwkv.modelForward(h, tk)
        want = m.forward(tk)[: mf.VOCAB]
        got = rwkv.getOutput(h)
        assert got.dtype == np.float32 and got.shape == (50277,) and np.array_equal(got, want)
    st = rwkv.getState(h)
    m.pull_state(1)
    assert len(st) == 5 and all(s.shape == (L * D,) for s in st)
    for a, b in zip(st, m.state.arrays()):
        assert np.array_equal(a, b[: L * D])
    assert 0 <= rwkv.typicalSample(h, 0.9, 0.8) < 50277
    rwkv.initState(h); rwkv.modelForward(h, 9)          # initState really resets the state forward() uses
    m.reset_state(); assert np.array_equal(rwkv.getOutput(h), m.forward(9)[: mf.VOCAB])
    rwkv.freeRwkv(h); m.close()


@pytest.mark.gpu
def test_pybind_module_two_models_side_by_side(built, tmp_path):
    """reference bindings/pybind/c_binding.cpp:28-33: every initRwkv() hands out its own RWKV -- two different models alive in
    one Python process through the module, forwards interleaved, one freed while the other goes on"""
    sys.path.insert(0, CSRC)
    import rwkv
    from rwkv_cpp_accelerated_amd import engine
    hs, ms = [], []
    for (L, D, seed) in ((2, 768, 81), (3, 1024, 82)):
        t = mf.synthetic_tensors(L, D, seed=seed)
        p = str(tmp_path / f"m{seed}.bin")
        mf.write_bin(p, L, D, t)
        h = rwkv.initRwkv()
        assert rwkv.loadModel(h, p) == (L, D)
        rwkv.initOutput(h); rwkv.initState(h)
        hs.append(h)
        m = engine.RWKV(resident=True); m.loadFile(p)
        ms.append(m)
    for tk in (5, 50000, 17):
        for h, m in zip(hs, ms):
            rwkv.modelForward(h, tk)
            assert np.array_equal(rwkv.getOutput(h), m.forward(tk)[: mf.VOCAB])
    rwkv.freeRwkv(hs[0])
    rwkv.modelForward(hs[1], 9)
    assert np.array_equal(rwkv.getOutput(hs[1]), ms[1].forward(9)[: mf.VOCAB])
    rwkv.freeRwkv(hs[1])
    for m in ms:
        m.close()
