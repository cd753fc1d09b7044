"""PIN of the sampler to the reference's OWN implementation, include/rwkv/sampler/typical.h:20-58 (NumCpp).

tests/golden/typical_ref.npz holds histograms of 20 000 draws of the reference's typical() per case (3 logits vectors x 5
(temp, tau) pairs), made by tools/make_typical_golden.py from oracle/_ref/libtypical_ref.so = the reference header compiled
where it lies.  Checked here, without a GPU:
  * include/rwkv_sampler.h typical_weights() (C++, the host twin of the device sampler): every token the reference drew
    has weight > 0, every token with a non-negligible weight was drawn, and the reference's frequencies fit the weights
    (chi-square); its own randomised typical() fits the same distribution.  FINDINGS pinned here: the reference's cut at
    tau never takes effect (typical.h:50 assigns into the temporary NumCpp's mask indexing returns) and nc::power takes an
    integer exponent (typical.h:52), so its draws follow softmax^uint8(1/temp) -- the recipe of its header comment
    (recipe=True) is REJECTED by the reference's own histograms;
  * the numpy recipe the GPU tests use (tests/test_sampler_gpu.py typical_weights) agrees with the C++ weights.
tests/test_sampler_gpu.py carries the pin to the device sampler (token-for-token vs the recipe, and its distribution
against the same golden histograms).  Where /root/reference exists a short live run of the reference is checked too."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sampler_recipe import sampler_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "typical_ref.npz")
V = 50277


def chi2_ok(counts, probs, n):
    """counts[i] observed of n draws, probs expected; bins with expectation < 5 are pooled.  Loose 5-sigma bound."""
    exp = probs * n
    big = exp >= 5
    o = np.append(counts[big], counts[~big].sum()); e = np.append(exp[big], exp[~big].sum())
    keep = e > 0
    o, e = o[keep], e[keep]
    dof = max(len(e) - 1, 1)
    chi2 = float(((o - e) ** 2 / e).sum())
    return chi2 <= dof + 5.0 * np.sqrt(2.0 * dof) + 5.0, chi2, dof


@pytest.fixture(scope="module")
def gold():
    assert os.path.exists(GOLD), "tests/golden/typical_ref.npz missing (tools/make_typical_golden.py)"
    return np.load(GOLD)


@pytest.fixture(scope="module")
def app(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("samp") / "sampler_app")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "sampler_app.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe])
    return exe


def cases(gold):
    for k in range(len(gold["scales"])):
        for j, (temp, tau) in enumerate(gold["pairs"]):
            yield k, j, float(temp), float(tau)


def test_host_sampler_vs_reference_typical_h(gold, app, tmp_path):
    n = int(gold["draws"])
    for k, j, temp, tau in cases(gold):
        logits = gold["logits"][k]
        lp, wp = str(tmp_path / "l.bin"), str(tmp_path / "w.f64")
        logits.tofile(lp)
        subprocess.check_call([app, lp, repr(temp), repr(tau), "--weights", wp])
        w = np.fromfile(wp, dtype=np.float64)
        ids, cnt = gold[f"ids_{k}_{j}"], gold[f"cnt_{k}_{j}"]
        assert cnt.sum() == n
        assert (w[ids] > 0).all(), f"case {k},{j}: the reference drew tokens outside our kept set"
        probs = w / w.sum()
        counts = np.zeros(V); counts[ids] = cnt
        missing = np.nonzero((probs * n > 30) & (counts == 0))[0]
        assert missing.size == 0, f"case {k},{j}: tokens {missing[:5]} have weight but the reference never drew them"
        ok, chi2, dof = chi2_ok(counts, probs, n)
        assert ok, f"case {k},{j} (temp {temp}, tau {tau}): chi2 {chi2:.1f} for {dof} dof"
        # the numpy recipe of the GPU tests is the same function
        wr = sampler_weights(logits, temp, tau, recipe=False)
        assert np.array_equal(wr > 0, w > 0) and np.allclose(wr, w, rtol=1e-11, atol=0)


def test_documented_cut_is_not_what_the_reference_does(gold):
    """with the cut of the header comment applied, peaked cases keep a single token -- the reference drew several"""
    k, j = 0, 0
    temp, tau = (float(x) for x in gold["pairs"][j])
    w = sampler_weights(gold["logits"][k], temp, tau, recipe=True)
    ids = gold[f"ids_{k}_{j}"]
    assert (w > 0).sum() < len(ids) and not (w[ids] > 0).all()


def test_host_randomised_typical_fits_the_reference_histogram(gold, app, tmp_path):
    """include/rwkv_sampler.h typical() (std::discrete_distribution) vs the reference's nc::random::discrete draws: two
    samples of the same distribution (two-sample chi-square on the pooled bins)"""
    k, j = 0, 4                      # storygen's temp 0.8 / tau 0.7
    temp, tau = (float(x) for x in gold["pairs"][j])
    lp = str(tmp_path / "l.bin"); gold["logits"][k].tofile(lp)
    m = 4000
    out = subprocess.run([app, lp, repr(temp), repr(tau), "--draw", str(m)], capture_output=True, text=True, env=dict(os.environ, RWKV_SAMPLER_SEED="5"))
    mine = np.bincount(np.array(out.stdout.split(), dtype=np.int64), minlength=V).astype(np.float64)
    ref = np.zeros(V); ref[gold[f"ids_{k}_{j}"]] = gold[f"cnt_{k}_{j}"]
    n = ref.sum()
    assert (ref[mine > 0] > 0).all() or (mine[(ref == 0)] .sum() < 0.002 * m)
    ok, chi2, dof = chi2_ok(mine, ref / n, m)
    assert ok, f"chi2 {chi2:.1f} for {dof} dof"


def test_live_reference_typical_h_when_present(gold):
    so = os.path.join(ROOT, "oracle", "_ref", "libtypical_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtypical_ref.so not built (needs /root/reference at build time)")
    L = C.CDLL(so)
    L.typical_ref_draw.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.typical_ref_seed.argtypes = [C.c_uint32]
    L.typical_ref_seed(123)
    for k, j in ((0, 0), (1, 2), (2, 3)):
        temp, tau = (float(x) for x in gold["pairs"][j])
        logits = np.ascontiguousarray(gold["logits"][k])
        out = np.zeros(150, np.int32)
        L.typical_ref_draw(logits.ctypes.data, temp, tau, len(out), out.ctypes.data)
        w = sampler_weights(logits, temp, tau, recipe=False)
        assert (w[out] > 0).all()
