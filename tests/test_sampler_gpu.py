"""Device-side sampler (csrc/sampler.hip.h) against the reference's sampler include/rwkv/sampler/typical.h:20-58: token
for token against tests/sampler_recipe.py (both modes: the reference as compiled = default, and the recipe its header
documents), and -- the pin to the reference's OWN code -- its distribution against the histograms of 20 000 draws of the
reference's typical() in tests/golden/typical_ref.npz (tools/make_typical_golden.py)."""
import os

import numpy as np
import pytest

from sampler_recipe import sampler_weights, sampler_u

from rwkv_cpp_accelerated_amd import modelfile as mf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod(built):
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from rwkv_cpp_accelerated_amd import engine
    engine.lib()
    return engine


def splitmix_u(seed, step):
    m = (1 << 64) - 1
    x = (seed + step + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return (x >> 11) / 9007199254740992.0


def _near_boundary(logits, temp, tau, u, ban0, recipe, eps=1e-6):
    """u lands within eps of a CDF step (or the kept set is decided within eps of tau): a legitimate tie"""
    l = np.array(logits, np.float32, copy=True)
    if ban0:
        l[0] = -99.0
    w = sampler_weights(l, temp, tau, recipe)
    c = np.cumsum(w) / w.sum()
    return np.abs(c - u).min() < eps


@pytest.mark.parametrize("recipe", [False, True])
@pytest.mark.parametrize("temp,tau", [(0.9, 0.8), (1.0, 0.95), (0.5, 0.2), (2.0, 0.999), (1.0, 1.5), (0.3, 0.5)])
def test_sample_matches_host_recipe(eng_mod, temp, tau, recipe):
    L, D = 2, 256
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, mf.synthetic_tensors(L, D, seed=11, head_scale=30.0))
    rng = np.random.default_rng(int(temp * 100 + tau * 1000))
    mismatches = 0
    for tk in (5, 77, 50000):
        logits = m.forward(tk)[: mf.VOCAB].copy()
        for ban0 in (False, True):
            for u in list(rng.random(12)) + [0.0, 0.999999999]:
                got = m.sample_typical(temp, tau, u, ban0=ban0, recipe=recipe)
                want = sampler_u(logits, temp, tau, u, recipe, ban0)
                if got != want and not _near_boundary(logits, temp, tau, u, ban0, recipe):
                    mismatches += 1
    assert mismatches == 0
    m.close()


@pytest.mark.parametrize("recipe", [False, True])
def test_decode_typical_equals_host_loop_and_is_reproducible(eng_mod, recipe):
    L, D, n = 2, 256, 24
    t = mf.synthetic_tensors(L, D, seed=12, head_scale=30.0)
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(L, D, t)
    ids = m.decode_typical(9, n, temp=0.45, tau=0.8, seed=1234, recipe=recipe)
    m.reset_state()
    ids2 = m.decode_typical(9, n, temp=0.45, tau=0.8, seed=1234, recipe=recipe)
    assert np.array_equal(ids, ids2)
    m.reset_state()
    other = m.decode_typical(9, n, temp=0.45, tau=0.8, seed=99, recipe=recipe)
    assert not np.array_equal(ids, other)
    # host loop: same engine logits, host recipe, same uniforms
    m.reset_state()
    tk, host = 9, []
    for step in range(n):
        logits = m.forward(tk)[: mf.VOCAB].copy()
        u = splitmix_u(1234, step)
        want = sampler_u(logits, 0.45, 0.8, u, recipe, ban0=True)
        if want != int(ids[step]):
            assert _near_boundary(logits, 0.45, 0.8, u, True, recipe), f"step {step}: device {ids[step]} host {want}"
            want = int(ids[step])          # follow the device past a legitimate tie
        host.append(want); tk = want
    assert host == [int(v) for v in ids]
    m.close()


def _logits_view(m):
    """the engine's device logits buffer (row 0) as a torch tensor, to plant a chosen logits vector"""
    import torch
    from rwkv_cpp_accelerated_amd import engine

    class _A:
        __cuda_array_interface__ = dict(shape=(mf.VOCAB,), typestr="<f4", data=(int(engine.lib().rwkv_logits_device(m._h)), False), version=2)
    return torch.as_tensor(_A(), device="cuda:0")


@pytest.mark.parametrize("k,j", [(0, 0), (1, 4), (2, 2), (0, 3)])
def test_device_sampler_fits_the_reference_histograms(eng_mod, k, j):
    """PIN to the reference's own typical(): the device sampler (default mode), driven with a stratified grid of uniforms on
    the golden logits vector, against the histogram of 20 000 draws of the reference (two-sample chi-square, pooled bins)"""
    import torch
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "typical_ref.npz"))
    temp, tau = (float(x) for x in gold["pairs"][j])
    m = eng_mod.RWKV(resident=True)
    m.loadTensors(2, 64, mf.synthetic_tensors(2, 64, seed=3))
    m.forward(5)
    _logits_view(m).copy_(torch.from_numpy(np.ascontiguousarray(gold["logits"][k])).cuda())
    torch.cuda.synchronize()
    M = 6000
    dev = np.zeros(mf.VOCAB)
    for i in range(M):
        dev[m.sample_typical(temp, tau, (i + 0.5) / M)] += 1
    ref = np.zeros(mf.VOCAB); ref[gold[f"ids_{k}_{j}"]] = gold[f"cnt_{k}_{j}"]
    n = ref.sum()
    pooled = (dev * (n / M) + ref) < 10                       # pool the bins with a small combined expectation
    a = np.append(dev[~pooled], dev[pooled].sum()); b = np.append(ref[~pooled], ref[pooled].sum())
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    chi2 = float((((a * np.sqrt(n / M) - b * np.sqrt(M / n)) ** 2) / (a + b)).sum())
    dof = max(len(a) - 1, 1)
    assert chi2 <= dof + 5.0 * np.sqrt(2.0 * dof) + 5.0, f"chi2 {chi2:.1f} for {dof} dof"
    # and every token the reference drew with some frequency is reachable on the device
    assert (dev[ref > 40] > 0).all()
    m.close()
