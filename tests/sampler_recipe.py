"""The two behaviours of the reference sampler (reference include/rwkv/sampler/typical.h:20-58), restated in numpy for the
sampler tests (CPU and GPU)."""
import numpy as np


def sampler_weights(logits, temp, tau, recipe):
    """recipe=False: what the reference's typical.h COMPUTES (no cut -- typical.h:50 assigns into a temporary -- and an
    integer exponent uint8(1/temp), typical.h:52 / NumCpp power.hpp:68); recipe=True: what its header comment documents"""
    l = logits.astype(np.float64)
    e = np.exp(l - l.max()); p = e / e.sum()
    if not recipe:
        if np.float32(temp) == np.float32(1.0):
            return p
        n = min(int(1.0 / float(np.float32(temp))), 255)
        return np.ones_like(p) if n == 0 else p ** n
    with np.errstate(divide="ignore", invalid="ignore"):
        nl = -np.log(p); ent = np.nansum(nl * p); sh = np.abs(nl - ent)
    ids = np.argsort(sh, kind="stable")
    cutoff = min(int((np.cumsum(p[ids]) < tau).sum()), len(p) - 1)
    w = np.where(sh > sh[ids[cutoff]], 0.0, p)
    return w ** (1.0 / temp) if temp != 1.0 else w


def sampler_u(logits, temp, tau, u, recipe, ban0=False):
    """inverse CDF in token order for the uniform u: the draw include/rwkv_sampler.h typical_u() and the device sampler make"""
    l = np.array(logits, np.float32, copy=True)
    if ban0:
        l[0] = -99.0
    w = sampler_weights(l, temp, tau, recipe)
    c = np.cumsum(w)
    i = int(np.searchsorted(c, u * c[-1], side="right"))
    nz = np.nonzero(w)[0]
    return int(nz[-1]) if i >= len(w) else int(i if w[i] > 0 else nz[nz > i][0])
