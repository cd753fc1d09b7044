"""Host-side restatement of how csrc/tile.hip.h's consumer waves feed v_mfma_i32_16x16x64_i8 (round 6): the fragment as it lies in the ring is the B
operand, the staged vector the A operand; for 4-row tiles the products that belong together are the DIAGONAL blocks of D.  The operand and accumulator
register images are the ones csrc/seq.hip.h documents for the chunk path (A: lane l = row l % 16, k-piece l / 16; B: lane l = column l % 16, k-piece
l / 16; D: lane l holds column l % 16, rows 4 (l / 16) + i in register i) -- on the GPU the bit-identity of tile form and row form checks the same thing
end to end (tests/test_engine_gpu.py); this test pins the index arithmetic where it can be read."""
import numpy as np


def mfma_16x16x64_i8(a_lanes, b_lanes, acc):
    """a_lanes, b_lanes: int8 [64 lanes][16 bytes]; acc: int64 [64 lanes][4 registers].  D[m][n] += sum_j sum_e A[m][j][e] * B[n][j][e]."""
    A = np.zeros((16, 64), np.int64); B = np.zeros((16, 64), np.int64)
    for l in range(64):
        A[l % 16, 16 * (l // 16):16 * (l // 16) + 16] = a_lanes[l]
        B[l % 16, 16 * (l // 16):16 * (l // 16) + 16] = b_lanes[l]
    D = A @ B.T                                            # [m][n]
    out = acc.copy()
    for l in range(64):
        for i in range(4):
            out[l, i] += D[4 * (l // 16) + i, l % 16]
    return out


def staged(xl, kb_pieces):
    """the staged vector's order: [16-byte piece along k][limb][16 bytes] (tile.hip.h stage_quad_t); xl: int8 [3][K]"""
    K = xl.shape[1]
    return np.stack([np.stack([xl[b, 16 * p:16 * p + 16] for b in range(3)]) for p in range(K // 16)])      # [piece][limb][16]


def test_a_16_row_fragment_is_the_b_operand_and_the_three_limbs_are_rows_0_to_2():
    rng = np.random.default_rng(1)
    K = 256                                                # four fragments of 64 inputs
    W = rng.integers(-128, 128, (16, K)).astype(np.int8)
    xl = rng.integers(-128, 128, (3, K)).astype(np.int8)
    xs = staged(xl, K // 16)
    acc = np.zeros((64, 4), np.int64)
    for f in range(K // 64):                               # fragment f: lane l = piece l / 16 of row l % 16 (k_bimage, TH = 16)
        b = np.stack([W[l % 16, 64 * f + 16 * (l // 16):64 * f + 16 * (l // 16) + 16] for l in range(64)])
        a = np.stack([xs[4 * f + l // 16, min(l % 16, 2)] for l in range(64)])          # tile_consume: piece pc = l / 16, limb r < 3 ? r : 2
        acc = mfma_16x16x64_i8(a, b, acc)
    want = xl.astype(np.int64) @ W.astype(np.int64).T      # [limb][row]
    for n in range(16):                                    # lanes 0..15 hold row n's limb sums in registers 0..2
        assert [acc[n, i] for i in range(3)] == list(want[:, n])


def test_a_4_row_fragment_is_sixteen_virtual_rows_and_the_diagonal_blocks_belong_together():
    rng = np.random.default_rng(2)
    K = 512                                                # two fragments of 256 inputs
    W = rng.integers(-128, 128, (4, K)).astype(np.int8)
    xl = rng.integers(-128, 128, (3, K)).astype(np.int8)
    xs = staged(xl, K // 16)
    acc = np.zeros((64, 4), np.int64)
    for f in range(K // 256):                              # fragment f: lane l = piece l / 4 (of 16) of row l % 4 (k_bimage, TH = 4)
        b = np.stack([W[l % 4, 256 * f + 16 * (l // 4):256 * f + 16 * (l // 4) + 16] for l in range(64)])
        a = []
        for l in range(64):                                # tile_consume: m = min(l % 16, 11) = (limb m / 4, q = m % 4); piece 4 (l / 16) + q
            m = min(l % 16, 11)
            a.append(xs[16 * f + 4 * (l // 16) + (m & 3), m >> 2])
        acc = mfma_16x16x64_i8(np.stack(a), b, acc)
    want = xl.astype(np.int64) @ W.astype(np.int64).T      # [limb][row]
    for g in range(3):                                     # lane (n, g): register n / 4; the four q of a row sit 4 lanes apart
        v = np.array([acc[16 * g + n, n // 4] for n in range(16)])
        for row in range(4):
            assert v[row] + v[row + 4] + v[row + 8] + v[row + 12] == want[g, row]
