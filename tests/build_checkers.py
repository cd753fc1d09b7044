"""Build the CHECKERS: oracle/ (the CPU restatement and -- only where /root/reference exists -- oracle/_ref, the reference's own
sources compiled where they lie) and tests/fake_rccl.cpp (the RCCL stand-in of the world > 1 tests).  Test infrastructure: nothing
under rwkv-cpp-accelerated_amd/ imports this file; callers are tests/conftest.py, __graft_entry__.build() / smoke() and bench.py's
cpu_baseline / reference-kernel legs."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from rwkv_cpp_accelerated_amd.build import ARCH, HIPCC, _run, _stale, _stamp, reference_root  # noqa: E402


def build_oracle(force: bool = False):
    """oracle/: the C restatement, and -- only where /root/reference exists -- oracle/_ref."""
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "librwkv_oracle.so")
    if force or _stale(so, [os.path.join(odir, "rwkv_oracle.c")]):
        _run(["make", "-C", odir, "-B", "librwkv_oracle.so"])
    ref_root = reference_root()
    ref_so = os.path.join(odir, "_ref", "libref.so")
    if ref_root:
        if force or _stale(ref_so, [os.path.join(odir, "ref_driver.cpp")]):
            _run(["make", "-C", odir, "ref", f"REF={ref_root}"])
        # the reference's sampler and the reference's own caller (storygen) built against the drop-in: checkers / evidence
        # that only the authoring container can compile (they read /root/reference at BUILD time, never at run time)
        _run(["make", "-C", odir, "typical", "storygen", "storygen_l2", "callers", f"REF={ref_root}", f"ROOT={ROOT}"])
    return so, (ref_so if os.path.exists(ref_so) else None)


def build_test_helpers(force: bool = False):
    """tests/fake_rccl.cpp -> tests/_build/libfake_rccl.so: the eight librccl entry points the pipeline transport resolves, over
    shared memory, so that the native schedule runs with world > 1 on a one-GPU box (RWKV_RCCL_LIB).  Test infrastructure."""
    src = os.path.join(ROOT, "tests", "fake_rccl.cpp")
    out = os.path.join(ROOT, "tests", "_build", "libfake_rccl.so")
    if not os.path.exists(src):
        return None
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or _stale(out, [src]):
        _run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-x", "hip", f"--offload-arch={ARCH}", src, "-o", out, "-lrt", "-lpthread"])
        _stamp(out, [src])
    return out


def build_all(force: bool = False):
    """product + checkers: what the test fixtures and __graft_entry__.build() want on disk"""
    from rwkv_cpp_accelerated_amd import build
    out = build.build_all(force)
    ora = build_oracle(force)
    out.update(oracle=ora[0], ref=ora[1], fake_rccl=build_test_helpers(force))
    return out
