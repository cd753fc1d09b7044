"""Pin the CPU oracle to the REFERENCE'S OWN KERNEL through committed fixtures (tests/golden/*.npz,
produced on the MI355X box by tools/make_golden.py from oracle/_ref = reference rwkv.cu + rwkv.h
built unmodified with hipcc).  Runs without a GPU."""
import glob
import os

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf
import parity

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_*.npz")))


def _replay(forward_chunk, g):
    """feed the fixture's token chunks through forward_chunk(tokens) -> [T][V] logits and compare"""
    pos = 0
    stride = int(g["stride"])
    for n in g["chunk_len"]:
        toks = [int(x) for x in g["tokens"][pos:pos + n]]
        lg = forward_chunk(toks)
        for i in range(n):
            ref_s = g["sample"][pos + i]; got_s = lg[i][::stride]
            scale = np.abs(ref_s).max()
            assert np.abs(got_s - ref_s).max() <= parity.REL * scale, f"step {pos + i}"
            idx = g["top_idx"][pos + i]
            assert np.abs(lg[i][idx] - g["top_val"][pos + i]).max() <= parity.REL * scale
            got_id = parity.argmax_ban0(lg[i])
            if got_id != int(g["argmax"][pos + i]):     # allowed only inside the tolerance band of the top-2 margin
                tv = np.sort(g["top_val"][pos + i]); assert tv[-1] - tv[-2] <= 2 * parity.REL * scale
        pos += n


@pytest.mark.skipif(not GOLDEN, reason="no golden fixtures committed yet")
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_fixture(oracle, path):
    g = np.load(path)
    L, D, seed, mode = int(g["L"]), int(g["D"]), int(g["seed"]), int(g["mode"])
    t = mf.synthetic_tensors(L, D, seed=seed)
    m = oracle.from_tensors(L, D, t)
    slots = int(g["chunk_len"].max()) if mode == 0 else 1
    st = m.new_state(slots=slots)
    _replay(lambda toks: m.forward(toks, st, mode=mode), g)
    for i in range(5):
        n = min(st[i].size, g[f"state{i}"].size)      # GPT mode: only slot 0 is live (the reference sizes its host state by maxGPT)
        r = g[f"state{i}"][:n]
        assert np.abs(st[i][:n] - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), f"state {i}"
    m.close()
