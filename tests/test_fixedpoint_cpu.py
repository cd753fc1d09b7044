"""Host-side checks of the fixed-point algebra the HIP kernels rely on (csrc/kernels.hip.h stage_quad /
row_value, csrc/seq.hip.h seq_store_quad / k_mm8_seq epilogue): pure numpy, no GPU, no oracle.

1. the magic-number quantiser: bytes 0..2 of float32(x * inv_s + 1.5 * 2^23) are the three limbs of
   rint(x * inv_s) + 2^22 (round-half-even), for |x * inv_s| <= QLIM;
2. the limb decomposition reproduces sum_j u_j q_j exactly, and the scale turns it back into sum_j u_j x_j
   within the quantisation bound;
3. the signed-operand identity of the MFMA path: sum u l = sum (u-128)(l-128) + 128 rowsum(u) + 128 sum(l-128),
   folded over limbs with SEQ_CU = 128 * 65793 - 2^22."""
import numpy as np

QLIM = np.float32(4194000.0)
QOFF = 4194304
QMAGIC = np.float32(12582912.0)
SEQ_CU = 4227200


def quantise_limbs(x, amax):
    inv_s = QLIM / np.float32(max(amax, 1e-30))
    # fma(x, inv_s, QMAGIC): the product of two f32 is exact in f64 and the sum fits f64, so one rounding to f32
    t = (x.astype(np.float64) * np.float64(inv_s) + np.float64(QMAGIC)).astype(np.float32)
    bits = t.view(np.uint32)
    return (bits & 0xFF).astype(np.int64), ((bits >> 8) & 0xFF).astype(np.int64), ((bits >> 16) & 0xFF).astype(np.int64), inv_s


def test_magic_number_quantiser_is_round_half_even_plus_offset():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 3, np.float32([0, 1e-30, -1e-30, 7.5, -7.5])])
    amax = float(np.abs(x).max())
    l0, l1, l2, inv_s = quantise_limbs(x, amax)
    q = np.rint(x.astype(np.float64) * np.float64(inv_s)).astype(np.int64)       # rint = round half to even
    assert np.abs(q).max() <= int(QLIM)
    assert np.array_equal(l0 + 256 * l1 + 65536 * l2, q + QOFF)
    # exact ties round to even, like v_cvt_rpi would NOT: x * inv_s = k + 0.5
    ties = (np.arange(-9, 10, dtype=np.float64) + 0.5)
    xt = (ties / np.float64(inv_s)).astype(np.float32)
    ok = (xt.astype(np.float64) * np.float64(inv_s)) == ties                      # keep the ones that are exactly representable
    a0, a1, a2, _ = quantise_limbs(xt[ok], amax)
    assert np.array_equal(a0 + 256 * a1 + 65536 * a2 - QOFF, np.rint(ties[ok]).astype(np.int64))


def test_limb_contraction_is_exact_and_scales_back():
    rng = np.random.default_rng(1)
    for N in (16, 768, 4096, 16384):
        x = (rng.standard_normal(N) * np.exp(rng.uniform(-3, 3, N))).astype(np.float32)   # heavy-tailed
        u = rng.integers(0, 256, N).astype(np.int64)
        amax = float(np.abs(x).max())
        l0, l1, l2, inv_s = quantise_limbs(x, amax)
        q = l0 + 256 * l1 + 65536 * l2 - QOFF
        T = (u * l0).sum() + 256 * (u * l1).sum() + 65536 * (u * l2).sum()       # what v_dot4_u32_u8 accumulates per limb
        assert (u * l0).sum() < 2 ** 32                                         # u32 accumulators cannot overflow at N <= 16384... (255*255*16384 < 2^32)
        assert T - QOFF * u.sum() == (u * q).sum()                              # row_value's integer part, exactly
        scale = np.float64(max(amax, 1e-30)) / np.float64(QLIM)
        got = scale * float(T - QOFF * u.sum())
        exact = float((u.astype(np.float64) * x.astype(np.float64)).sum())
        assert abs(got - exact) <= 0.5 * scale * u.sum() * (1 + 1e-6) + 1e-12   # half a quantisation step per element, worst case


def test_signed_operand_identity_of_the_mfma_path():
    rng = np.random.default_rng(2)
    for K in (64, 4096, 16384):
        x = rng.standard_normal(K).astype(np.float32)
        u = rng.integers(0, 256, K).astype(np.int64)
        amax = float(np.abs(x).max())
        limbs = quantise_limbs(x, amax)[:3]
        ub = u - 128                                     # weight byte ^ 0x80 reinterpreted as int8
        M, sa = [], []
        for l in limbs:
            a = l - 128                                  # stored A-operand byte
            assert a.min() >= -128 and a.max() <= 127 and ub.min() >= -128 and ub.max() <= 127
            M.append(int((ub * a).sum()))                # v_mfma_i32_16x16x64_i8 accumulator
            assert abs(M[-1]) < 2 ** 31
            sa.append(int(a.sum()))
            assert int((u * l).sum()) == M[-1] + 128 * int(u.sum()) + 128 * sa[-1]
        cA = sum(128 * (256 ** b) * sa[b] for b in range(3))                     # seq_finish
        lhs = M[0] + 256 * M[1] + 65536 * M[2] + cA + SEQ_CU * int(u.sum())      # k_mm8_seq epilogue (before the scale)
        q = limbs[0] + 256 * limbs[1] + 65536 * limbs[2] - QOFF
        assert lhs == int((u * q).sum())
