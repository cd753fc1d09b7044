"""Build the HIP engine for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
ENGINE_LIBS = []   # extra link flags of librwkv_mi355x.so


def _digest(sources, extra="") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for s in sources:
        if os.path.exists(s):
            h.update(os.path.relpath(s, ROOT).encode())      # not the absolute path: the tree is copied to another root on the GPU box
            with open(s, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _stale(target: str, sources, extra="") -> bool:
    """content-based: `<target>.stamp` holds the digest of the sources the target was built from (mtimes do not survive
    the copy to the GPU box, and a rebuild there would lose what only the authoring container can compile in)"""
    if not os.path.exists(target):
        return True
    stamp = target + ".stamp"
    if not os.path.exists(stamp):
        t = os.path.getmtime(target)
        if any(os.path.getmtime(s) > t for s in sources if os.path.exists(s)):
            return True
        _stamp(target, sources, extra)      # fresh by mtime (authoring container): record the digest for the copies of this tree
        return False
    with open(stamp) as f:
        return f.read().strip() != _digest(sources, extra)


def _stamp(target: str, sources, extra=""):
    with open(target + ".stamp", "w") as f:
        f.write(_digest(sources, extra))


def reference_root():
    r = os.environ.get("RWKV_REFERENCE", "/root/reference")
    return r if os.path.isdir(os.path.join(r, "include", "rwkv")) else None


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_engine(force: bool = False) -> str:
    """csrc/engine.hip (+ kernels.hip.h, seq.hip.h) -> csrc/librwkv_mi355x.so"""
    out = os.path.join(CSRC, "librwkv_mi355x.so")
    srcs = [os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "kernels.hip.h"), os.path.join(CSRC, "seq.hip.h"), os.path.join(CSRC, "sampler.hip.h"),
            os.path.join(ROOT, "include", "rwkv_mi355x.h")]
    if force or _stale(out, srcs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-Wno-unused-result", os.path.join(CSRC, "engine.hip"), "-o", out] + ENGINE_LIBS)
        _stamp(out, srcs)
    return out


def build_pybind(force: bool = False):
    """csrc/pybind_module.cpp -> csrc/rwkv.<abi>.so : the reference's pybind module `rwkv`
    (bindings/pybind/c_binding.cpp:158-175) on top of include/rwkv.h.  Optional: skipped when the
    source is not there yet."""
    src = os.path.join(CSRC, "pybind_module.cpp")
    if not os.path.exists(src):
        return None
    import sysconfig
    import pybind11
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(CSRC, "rwkv" + ext)
    hdrs = [os.path.join(ROOT, "include", f) for f in ("rwkv.h", "rwkv_mi355x.h", "rwkv_sampler.h")]
    # With the reference's include directory on the path include/rwkv.h pulls in the reference's own tokenizer and the module
    # gains initTokenizer / tokenizerEncode / tokenizerDecode (c_binding.cpp:158-175): the upstream headers are a BUILD
    # dependency of the full module surface.  A stale module is always rebuilt (a stale one next to a rebuilt engine would call
    # the C-ABI with an old argument list); where the headers are absent that rebuild would silently drop the three tokenizer
    # forwards, so it is refused unless asked for (RWKV_PYBIND_NO_TOKENIZER=1).
    ref = reference_root()
    if force or _stale(out, [src] + hdrs):
        if not ref and os.environ.get("RWKV_PYBIND_NO_TOKENIZER") != "1":
            raise RuntimeError(f"{out} is stale and the reference's tokenizer headers (RWKV_REFERENCE, default /root/reference) are not here: "
                               "rebuild in the authoring container, or set RWKV_PYBIND_NO_TOKENIZER=1 to build the module without "
                               "initTokenizer / tokenizerEncode / tokenizerDecode")
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-I" + os.path.join(ROOT, "include")] +
             (["-I" + os.path.join(ref, "include")] if ref else []) +
             ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], src, "-o", out,
              "-L" + CSRC, "-lrwkv_mi355x", "-Wl,-rpath,$ORIGIN"])
        _stamp(out, [src] + hdrs)
    return out


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
    eng = build_engine(force)
    pyb = build_pybind(force)
    ora = build_oracle(force)
    fake = build_test_helpers(force)
    return dict(engine=eng, pybind=pyb, oracle=ora[0], ref=ora[1], fake_rccl=fake)
