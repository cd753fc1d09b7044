"""Which RCCL the engine's pipeline transport binds (rwkv_pipe_rccl_path; no GPU involved).

A PyTorch-ROCm process already carries torch/lib/librccl.so, built against the HIP runtime torch brought along; the engine's streams
and buffers live in that same runtime, so the transport must bind THAT copy and not load the system's RCCL of another ROCm release
beside it.  RWKV_RCCL_LIB overrides (the shared-memory stand-in of the multi-rank tests)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = "import sys; sys.path.insert(0, %r)\n%s\nfrom rwkv_cpp_accelerated_amd import engine\nprint('RCCL=' + engine.RWKV.pipe_rccl_path())"


def _probe(pre, env_extra=None):
    env = dict(os.environ)
    env.pop("RWKV_RCCL_LIB", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", PROBE % (ROOT, pre)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("RCCL=")][-1][5:]


def test_binds_the_rccl_torch_already_loaded():
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(bundled):
        pytest.skip("this torch build does not bundle librccl.so")
    got = _probe("import torch")
    assert os.path.realpath(got) == os.path.realpath(bundled), got


def test_env_override_wins():
    fake = os.path.join(ROOT, "tests", "_build", "libfake_rccl.so")
    if not os.path.exists(fake):
        pytest.skip("tests/_build/libfake_rccl.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    got = _probe("import torch", {"RWKV_RCCL_LIB": fake})
    assert os.path.realpath(got) == os.path.realpath(fake), got
