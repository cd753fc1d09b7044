"""Host sampler (include/rwkv_sampler.h: typical_weights / typical_u / typical, the mirror of reference
include/rwkv/sampler/typical.h:20-58) against the python recipe quoted in the reference's header comment,
restated in numpy.  No GPU: the header only needs the C++ standard library."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 50277


def typical_u_numpy(logits, temp, tau, u):
    l = logits.astype(np.float64)
    e = np.exp(l - l.max()); p = e / e.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        nl = -np.log(p); ent = np.nansum(nl * p); sh = np.abs(nl - ent)
    ids = np.argsort(sh, kind="stable")
    cutoff = min(int((np.cumsum(p[ids]) < tau).sum()), V - 1)
    w = np.where(sh > sh[ids[cutoff]], 0.0, p)
    if temp != 1.0:
        w = w ** (1.0 / temp)
    c = np.cumsum(w)
    i = int(np.searchsorted(c, u * c[-1], side="right"))
    nz = np.nonzero(w)[0]
    return int(nz[-1]) if i >= V else int(i if w[i] > 0 else nz[nz > i][0])


@pytest.fixture(scope="module")
def app(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("samp") / "sampler_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "sampler_app.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe])
    return exe


@pytest.mark.parametrize("scale,temp,tau", [(1.0, 0.9, 0.8), (4.0, 1.0, 0.95), (8.0, 0.5, 0.3), (0.2, 2.0, 0.999)])
def test_typical_u_matches_numpy_recipe(app, tmp_path, scale, temp, tau):
    rng = np.random.default_rng(int(scale * 10 + temp * 100))
    logits = (rng.standard_normal(V) * scale).astype(np.float32)
    logits[rng.integers(0, V, 5)] += 6.0 * scale            # a few clear favourites
    path = str(tmp_path / "logits.bin")
    logits.tofile(path)
    us = [0.0, 0.25, 0.5, 0.75, 0.999999] + [float(x) for x in rng.random(8)]
    out = subprocess.run([app, path, repr(temp), repr(tau)] + [repr(u) for u in us], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [int(x) for x in out.stdout.split()]
    want = [typical_u_numpy(logits, temp, tau, u) for u in us]
    assert got == want
