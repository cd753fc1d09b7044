"""Device-side typical sampler (csrc/sampler.hip.h) against the host recipe (reference
include/rwkv/sampler/typical.h:20-58, restated in numpy below and in include/rwkv_sampler.h typical_u)."""
import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from rwkv_cpp_accelerated_amd import engine
    engine.lib()
    return engine


def typical_weights(logits, temp, tau, ban0=False):
    l = logits.astype(np.float64).copy()
    if ban0:
        l[0] = -99.0
    e = np.exp(l - l.max())
    p = e / e.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        nl = -np.log(p)
        ent = np.nansum(nl * p)
        sh = np.abs(nl - ent)
    ids = np.argsort(sh, kind="stable")
    cum = np.cumsum(p[ids])
    cutoff = min(int((cum < tau).sum()), len(p) - 1)
    w = np.where(sh > sh[ids[cutoff]], 0.0, p)
    if temp != 1.0:
        w = w ** (1.0 / temp)
    return w


def typical_u(logits, temp, tau, u, ban0=False):
    w = typical_weights(logits, temp, tau, ban0)
    c = np.cumsum(w)
    i = int(np.searchsorted(c, u * c[-1], side="right"))
    nz = np.nonzero(w)[0]
    return int(nz[-1]) if i >= len(w) else int(i if w[i] > 0 else nz[nz > i][0])


def splitmix_u(seed, step):
    m = (1 << 64) - 1
    x = (seed + step + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return (x >> 11) / 9007199254740992.0


def _near_boundary(logits, temp, tau, u, ban0, eps=1e-6):
    """u lands within eps of a CDF step (or the kept set is decided within eps of tau): a legitimate tie"""
    w = typical_weights(logits, temp, tau, ban0)
    c = np.cumsum(w) / w.sum()
    return np.abs(c - u).min() < eps


@pytest.mark.parametrize("temp,tau", [(0.9, 0.8), (1.0, 0.95), (0.5, 0.2), (2.0, 0.999), (1.0, 1.5)])
def test_sample_matches_host_recipe(eng_mod, temp, tau):
    L, D = 2, 256
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, mf.synthetic_tensors(L, D, seed=11, head_scale=30.0))
    rng = np.random.default_rng(int(temp * 100 + tau * 1000))
    mismatches = 0
    for tk in (5, 77, 50000):
        logits = m.forward(tk)[: mf.VOCAB].copy()
        for ban0 in (False, True):
            for u in list(rng.random(12)) + [0.0, 0.999999999]:
                got = m.sample_typical(temp, tau, u, ban0=ban0)
                want = typical_u(logits, temp, tau, u, ban0)
                if got != want and not _near_boundary(logits, temp, tau, u, ban0):
                    mismatches += 1
    assert mismatches == 0
    m.close()


def test_decode_typical_equals_host_loop_and_is_reproducible(eng_mod):
    L, D, n = 2, 256, 24
    t = mf.synthetic_tensors(L, D, seed=12, head_scale=30.0)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t)
    ids = m.decode_typical(9, n, temp=0.9, tau=0.8, seed=1234)
    m.reset_state()
    ids2 = m.decode_typical(9, n, temp=0.9, tau=0.8, seed=1234)
    assert np.array_equal(ids, ids2)
    m.reset_state()
    other = m.decode_typical(9, n, temp=0.9, tau=0.8, seed=99)
    assert not np.array_equal(ids, other)
    # host loop: same engine logits, host recipe, same uniforms
    m.reset_state()
    tk, host = 9, []
    for step in range(n):
        logits = m.forward(tk)[: mf.VOCAB].copy()
        u = splitmix_u(1234, step)
        want = typical_u(logits, 0.9, 0.8, u, ban0=True)
        if want != int(ids[step]):
            assert _near_boundary(logits, 0.9, 0.8, u, True), f"step {step}: device {ids[step]} host {want}"
            want = int(ids[step])          # follow the device past a legitimate tie
        host.append(want); tk = want
    assert host == [int(v) for v in ids]
    m.close()
