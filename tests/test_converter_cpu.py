"""The numpy converter (rwkv_cpp_accelerated_amd.converter) against the REFERENCE converter's output
(fixtures tests/golden/converter_L2_D64.npz and converter_169M.npz -- BASELINE config 1's shape -- produced by
tools/make_converter_golden.py running /root/reference/converter/convert_model.py + cpp_save_tensor.cpp on the same
seeded checkpoint)."""
import hashlib
import os

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import converter, modelfile as mf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = os.path.join(GOLD, "converter_L2_D64.npz")
OFFSET_SLOTS = (mf.O1, mf.O2, mf.O3, mf.ATTOUTO, mf.FFNKO, mf.FFNVO, mf.FFNRO, mf.HEADO)


def converted_equals_reference_file(g, t):
    """every tensor of our conversion against the reference converter's file: sha256-identical, except that the offset
    vectors may differ by the rounding of torch's vs numpy's mean of the truncation residue (<= 4e-7 relative)"""
    n_exact = 0
    for i in range(mf.N_TENSORS):
        arr = np.ascontiguousarray(np.asarray(t[i], dtype=mf.DTYPES[i]).reshape(-1))
        same = hashlib.sha256(arr.tobytes()).hexdigest() == str(g[f"sha_{i}"])
        n_exact += same
        if not same:
            assert i in OFFSET_SLOTS, mf.NAMES[i]
            ref = g[f"val_{i}"]
            assert np.abs(arr - ref).max() <= 4e-7 * max(1e-3, np.abs(ref).max())
    assert n_exact >= mf.N_TENSORS - len(OFFSET_SLOTS)
    return n_exact


@pytest.mark.parametrize("name", ["L2_D64", "169M"])
def test_converter_matches_reference_converter(tmp_path, oracle, name):
    fix = os.path.join(GOLD, f"converter_{name}.npz")
    if not os.path.exists(fix):
        pytest.skip("converter fixture missing")
    g = np.load(fix)
    L, D, seed = int(g["L"]), int(g["D"]), int(g["seed"])
    w = converter.synthetic_state_dict(L, D, seed)
    l2, d2, t = converter.convert_state_dict(w)
    assert (l2, d2) == (L, D)
    if name == "L2_D64":
        p = str(tmp_path / "model.bin")
        mf.write_bin(p, L, D, t)
        assert os.path.getsize(p) == int(g["file_bytes"])
    else:
        assert mf.file_bytes(L, D) == int(g["file_bytes"])
    converted_equals_reference_file(g, t)
    # the C oracle's quantiser (restating convert_model.py:108-119) agrees with the converter on u8 and scale
    W = w["blocks.0.att.key.weight"]
    q, r, o = converter.quantize_matrix(W)
    q2, r2, o2 = oracle.quantize_matrix(W)
    assert np.array_equal(q, q2) and np.array_equal(r, r2) and np.abs(o - o2).max() <= 4e-7 * np.abs(o).max()


def test_converter_output_runs_in_oracle(tmp_path, oracle):
    """a converted checkpoint is a loadable model: dequantised weights reproduce the float model's first-layer key"""
    L, D = 1, 64
    w = converter.synthetic_state_dict(L, D, 7)
    _, _, t = converter.convert_state_dict(w)
    km = np.asarray(t[mf.KM]).reshape(D, D).astype(np.float64); kr = np.asarray(t[mf.KR]); ko = np.asarray(t[mf.O1])
    deq = km * kr[:, None] + ko[:, None]                      # [in][out]
    assert np.abs(deq - w["blocks.0.att.key.weight"].T).max() <= 1.01 * kr.max()
    m = oracle.from_tensors(L, D, t)
    lg = m.forward([5], m.new_state())
    assert np.isfinite(lg).all()
    m.close()
