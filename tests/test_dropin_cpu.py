"""Drop-in claims that need no GPU: the reference's OWN callers compile and link against this repo's boundary.

  * reference examples/storygen/storygen.cpp, from where it lies, against include/rwkv.h (level 1: drop-in header) and
    against the reference's own include/rwkv/rwkv/rwkv.h + integration/rwkv_backend_mi355x.cpp (level 2: backend TU
    implementing the prototypes rwkv.h:63-122 on the C-ABI) -- oracle/Makefile targets storygen / storygen_l2;
  * reference bindings/pybind/binding.py (TokenizerWrapper flow of reference tests/test_pybind.py) on the pybind module
    `rwkv` built from csrc/pybind_module.cpp.
They read /root/reference (build time only), so they skip where it is absent; the binaries travel to the GPU box where
tests/test_dropin_gpu.py RUNS them."""
import importlib
import os
import subprocess
import sys

import pytest

from rwkv_cpp_accelerated_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rwkv-cpp-accelerated_amd", "csrc")
REF = build.reference_root()
needs_ref = pytest.mark.skipif(REF is None, reason="/root/reference not present (authoring container only)")


@needs_ref
@pytest.mark.parametrize("exe", ["storygen_mi355x", "storygen_l2"])
def test_reference_storygen_builds_against_the_dropin(built, exe):
    p = os.path.join(ROOT, "oracle", "_ref", exe)
    assert os.path.exists(p), "oracle/Makefile did not produce it"
    # it really is linked against the engine's C-ABI (and nothing CUDA-ish)
    out = subprocess.run(["ldd", p], capture_output=True, text=True).stdout
    assert "librwkv_mi355x.so" in out and "cuda" not in out.lower()
    syms = subprocess.run(["nm", "-D", "--undefined-only", p], capture_output=True, text=True).stdout
    for s in ("rwkv_load_file", "rwkv_forward", "rwkv_set_state", "rwkv_get_output"):
        assert s in syms, s


@needs_ref
@pytest.mark.parametrize("exe", ["vectordb_mi355x", "vectordb_l2", "terminalchat_mi355x", "terminalchat_l2", "two_models_mi355x", "two_models_l2"])
def test_the_other_reference_callers_build_against_the_dropin(built, exe):
    """reference examples/vectordb/vectordb.cpp (the only user of loadFile(.., 5), emptyState(), raw state->statedd,
    vectordb.cpp:15-59) and examples/terminalchat/chat.cpp (chat.cpp:50-85), from where they lie, at both boundary levels; and
    the two-models test program (tests/cpp/two_models_app.cpp).  oracle/Makefile target `callers`."""
    p = os.path.join(ROOT, "oracle", "_ref", exe)
    assert os.path.exists(p), "oracle/Makefile did not produce it"
    out = subprocess.run(["ldd", p], capture_output=True, text=True).stdout
    assert "librwkv_mi355x.so" in out and "cuda" not in out.lower()
    assert "not found" not in out, out
    syms = subprocess.run(["nm", "-D", "--undefined-only", p], capture_output=True, text=True).stdout
    for s in ("rwkv_load_file", "rwkv_forward", "rwkv_set_state", "rwkv_get_output", "rwkv_free"):
        assert s in syms, s


@needs_ref
def test_backend_tu_defines_every_prototype_of_the_reference_header(built, tmp_path):
    """integration/rwkv_backend_mi355x.cpp vs the declarations in the reference's rwkv.h:63-122: a TU that includes the
    reference header and takes the address of each declared function with its exact type must link against it"""
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "rwkv/rwkv/rwkv.h"\n'
                     "int main() {\n"
                     "  auto a = &load; auto b = &setState; auto c = &getOutput; auto d = &freeTensors;\n"
                     "  auto e = &cuda_rwkv; auto f = &cuda_rwkv_parralel;\n"
                     "  return (a && b && c && d && e && f) ? 0 : 1;\n}\n")
    exe = str(tmp_path / "probe")
    subprocess.check_call(["g++", "-std=c++17", "-w", str(probe), os.path.join(ROOT, "integration", "rwkv_backend_mi355x.cpp"),
                           "-I" + os.path.join(REF, "include"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + CSRC, "-lrwkv_mi355x", "-Wl,-rpath," + CSRC, "-o", exe])


@needs_ref
def test_reference_binding_py_tokenizer_flow(built):
    """reference tests/test_pybind.py:18-27 with the reference's own bindings/pybind/binding.py, SO_LIB_PATH -> our module"""
    assert built["pybind"]
    sys.path.insert(0, CSRC)
    sys.path.insert(0, os.path.join(REF, "bindings", "pybind"))
    os.environ["SO_LIB_PATH"] = "rwkv"
    try:
        binding = importlib.import_module("binding")
        vocab = os.path.join(REF, "include", "rwkv", "tokenizer", "vocab")
        tok = binding.TokenizerWrapper(vocab_path=os.path.join(vocab, "vocab.json"), merges_path=os.path.join(vocab, "merges.txt"))
        ids = tok.encode("To see the world in a grain of")
        assert isinstance(ids, list) and all(isinstance(i, int) and 0 <= i < 50277 for i in ids) and len(ids) == 8
        assert "".join(tok.decode(i) for i in ids) == "To see the world in a grain of"
        for f in ("ModelWrapper", "TokenizerWrapper"):
            assert hasattr(binding, f)
        with pytest.raises(ValueError):
            binding.TokenizerWrapper(vocab_path="/nonexistent/vocab.json", merges_path="/nonexistent/merges.txt")
    finally:
        sys.path.remove(os.path.join(REF, "bindings", "pybind"))
        sys.modules.pop("binding", None)
