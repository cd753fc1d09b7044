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


def engine_sources():
    """every file librwkv_mi355x.so is compiled from: the staleness check and the `.stamp` digest (which decides whether the GPU box
    rebuilds) cover all of them -- csrc/*.hip, csrc/*.hip.h and the C-ABI header"""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hip.h"))) + [os.path.join(ROOT, "include", "rwkv_mi355x.h")]


def build_engine(force: bool = False) -> str:
    """csrc/engine.hip (+ every header it includes) -> csrc/librwkv_mi355x.so"""
    out = os.path.join(CSRC, "librwkv_mi355x.so")
    srcs = engine_sources()
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


def build_all(force: bool = False):
    """the PRODUCT's artefacts only: the engine and the pybind module.  The oracle, the reference build and the RCCL stand-in are test
    infrastructure and are built by tests/build_checkers.py (which __graft_entry__.build() and the test fixtures call)."""
    return dict(engine=build_engine(force), pybind=build_pybind(force))
