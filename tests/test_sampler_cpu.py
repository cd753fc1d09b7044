"""Host sampler (include/rwkv_sampler.h: typical_weights / typical_u / typical, the mirror of reference
include/rwkv/sampler/typical.h:20-58) against tests/sampler_recipe.py in both modes: the reference as compiled (pinned to
the reference's own draws by tests/test_sampler_ref_cpu.py) and the python recipe quoted in the reference's header comment.
No GPU: the header only needs the C++ standard library."""
import os
import subprocess

import numpy as np
import pytest

from sampler_recipe import sampler_u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 50277


@pytest.fixture(scope="module")
def app(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("samp") / "sampler_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "sampler_app.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe])
    return exe


@pytest.mark.parametrize("truncate", [False, True])     # False: the reference as compiled; True: the documented recipe
@pytest.mark.parametrize("scale,temp,tau", [(1.0, 0.9, 0.8), (4.0, 1.0, 0.95), (8.0, 0.5, 0.3), (0.2, 2.0, 0.999)])
def test_typical_u_matches_numpy_recipe(app, tmp_path, scale, temp, tau, truncate):
    rng = np.random.default_rng(int(scale * 10 + temp * 100))
    logits = (rng.standard_normal(V) * scale).astype(np.float32)
    logits[rng.integers(0, V, 5)] += 6.0 * scale            # a few clear favourites
    path = str(tmp_path / "logits.bin")
    logits.tofile(path)
    us = [0.0, 0.25, 0.5, 0.75, 0.999999] + [float(x) for x in rng.random(8)]
    out = subprocess.run([app, path, repr(temp), repr(tau)] + [repr(u) for u in us], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, RWKV_APP_TRUNCATE="1" if truncate else "0"))
    assert out.returncode == 0, out.stderr
    got = [int(x) for x in out.stdout.split()]
    want = [sampler_u(logits, temp, tau, u, truncate) for u in us]
    assert got == want
