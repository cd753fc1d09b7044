#!/usr/bin/env python3
"""Golden draws of the reference's OWN sampler (reference include/rwkv/sampler/typical.h:20-58, NumCpp) for
tests/test_sampler_ref_cpu.py and tests/test_sampler_gpu.py: tests/golden/typical_ref.npz.

Needs /root/reference (builds oracle/_ref/libtypical_ref.so through oracle/Makefile).  ~7 ms per draw: the cases run in
parallel processes.  usage: python tools/make_typical_golden.py [draws_per_case=20000]"""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 50277
PAIRS = [(0.9, 0.8), (1.0, 0.95), (0.5, 0.2), (2.0, 0.999), (0.8, 0.7)]   # (temp, tau); the last one is storygen's (storygen.cpp:67)
SCALES = [4.0, 8.0, 0.5]                                                    # peaked ... flat logits


def logits_case(k):
    rng = np.random.default_rng(1000 + k)
    l = (rng.standard_normal(V) * SCALES[k]).astype(np.float32)
    l[rng.integers(0, V, 5)] += 6.0 * SCALES[k]      # a few clear favourites
    return l


def work(job):
    k, j, n = job
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtypical_ref.so"))
    L.typical_ref_draw.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.typical_ref_seed.argtypes = [C.c_uint32]
    lg = logits_case(k)
    out = np.zeros(n, np.int32)
    L.typical_ref_seed(77 + 10 * k + j)
    L.typical_ref_draw(lg.ctypes.data, PAIRS[j][0], PAIRS[j][1], n, out.ctypes.data)
    ids, cnt = np.unique(out, return_counts=True)
    return k, j, ids.astype(np.int32), cnt.astype(np.int32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "typical"])
    jobs = [(k, j, n) for k in range(len(SCALES)) for j in range(len(PAIRS))]
    with mp.Pool(min(len(jobs), os.cpu_count() or 1)) as pool:
        res = pool.map(work, jobs)
    out = dict(pairs=np.array(PAIRS, np.float64), scales=np.array(SCALES), draws=np.array(n),
               logits=np.stack([logits_case(k) for k in range(len(SCALES))]))
    for k, j, ids, cnt in res:
        out[f"ids_{k}_{j}"] = ids
        out[f"cnt_{k}_{j}"] = cnt
    path = os.path.join(ROOT, "tests", "golden", "typical_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
