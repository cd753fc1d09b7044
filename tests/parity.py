"""shared parity helpers.  Tolerance = BASELINE.json north_star: logits within 1e-3 relative fp32
(max|d| <= 1e-3 * max|ref| and element-wise |d| <= 1e-3*|ref| + 1e-3*rms(ref)),
greedy ids identical wherever the reference's own top-2 margin exceeds that band."""
import numpy as np

REL = 1e-3


def check_logits(got, ref, what=""):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite logits"
    d = np.abs(got - ref)
    mx = np.abs(ref).max(); rms = np.sqrt((ref ** 2).mean())
    assert d.max() <= REL * mx, f"{what}: max|d|={d.max():.3e} > {REL}*max|ref|={REL * mx:.3e}"
    bad = d > REL * np.abs(ref) + REL * rms
    assert not bad.any(), f"{what}: {bad.sum()} elements outside 1e-3*|ref| + 1e-3*rms"
    return d.max() / mx


def argmax_ban0(logits):
    l = np.array(logits, np.float32, copy=True); l[0] = -np.inf
    return int(np.argmax(l))


def check_argmax(got, ref, what=""):
    """identical greedy id, unless the reference's top-2 margin is inside the tolerance band"""
    g, r = argmax_ban0(got), argmax_ban0(ref)
    if g != r:
        rr = np.array(ref, np.float64); rr[0] = -np.inf
        top2 = np.sort(rr)[-2:]
        margin = top2[1] - top2[0]
        assert margin <= 2 * REL * np.abs(rr[np.isfinite(rr)]).max(), f"{what}: argmax {g} != {r} with margin {margin:.3e}"
    return g == r
