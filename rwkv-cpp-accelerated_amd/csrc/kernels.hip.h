// kernels.hip.h -- gfx950 (MI355X / CDNA4) device code of the RWKV-v4 uint8 engine.
//
// One token = 4 launches per layer + embed + head (+ argmax), replacing the ~28 stream
// operations per layer of the reference driver (rwkv.cu:528-581):
//
//   k_first       rwkv.cu:513-524   embedding row gather (device-resident table) + ln0; opens ln1 site
//   k_att         rwkv.cu:535-545   ln1 + token-shift mix + K/V/R dequant-GEMV + WKV recurrence
//   k_attout      rwkv.cu:548-553   att_out dequant-GEMV + residual; commits state xy; opens ln2 site
//   k_ffn_rk      rwkv.cu:557-573   ln2 + mix + ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2
//   k_ffnv        rwkv.cu:574-577   ffn_v dequant-GEMV, x += v * sigmoid(r); commits state dd; opens next site
//   k_head        rwkv.cu:585-589   ln_out + head dequant-GEMV -> logits (+ per-block argmax)
//
// Design (DESIGN.md has the long form and the measurements behind each choice):
//  * The uint8 matrices are re-tiled at load into ROW-PER-OUTPUT order (w_t[k][j], j contiguous),
//    so one wavefront owns whole output rows: each lane streams 16-byte pieces of the row with
//    non-temporal global_load_dwordx4 (1 KiB per wave-instruction, every 128-B line used once)
//    and the row is finished by a 64-lane shuffle reduction.  No float atomics, no pre-zeroed
//    accumulators, no cross-workgroup reduction: results are deterministic and every epilogue
//    (WKV, sigmoid, relu^2, residual) fuses into the kernel that finished the row.
//  * sum_j x_j (u_jk r_j + o_j) = sum_j (x_j r_j) u_jk + sum_j x_j o_j.  The first term is an
//    integer contraction: the pre-scaled activation vector is quantised ONCE per workgroup to
//    23-bit fixed point (3 unsigned byte limbs, scale = max|.|/4194000) and staged in LDS, and
//    the u8 x u8 products run on v_dot4_u32_u8 (4 MACs per lane-instruction, exact u32
//    accumulation, no byte->float converts).  Measured on MI355X the f32 formulation
//    (v_cvt_f32_ubyteN + v_fma_f32 per weight byte) cost as much VALU time as the HBM stream
//    itself; the limb form needs 12 instructions per 16 weight bytes instead of 32 and its
//    rounding error (<= 1.2e-7 max|x| per element) is below that of an f32 FMA chain.  The offset term is
//    one scalar per vector.
//  * Every GEMV needs its complete input vector in every workgroup.  The LayerNorm / token-shift
//    work is split by ownership ("LayerNorm sites" below): the kernel that owns rows of x emits the
//    state-dependent part and per-workgroup partial sums, the consumer needs one reduction round
//    and one multiply-add per element before it can stage the vector.
//  * Grid = one 512-thread workgroup per CU (8 waves); each wave keeps R*S x 16 B loads in flight
//    and refills a step's registers with the next row group as soon as the step is consumed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkvk {

constexpr int NT = 512;          // threads per workgroup
constexpr int NW = NT / 64;      // wavefronts per workgroup
constexpr int RED_BYTES = 1024;  // LDS scratch for workgroup reductions
// rows [block_lo(n), block_hi(n)) of n belong to this workgroup: floor(b n / blocks) in 32-bit arithmetic (b n < 2^32 for every
// use: blocks <= 512, n <= 65536).  As a 64-bit expression it compiled to two ~130-instruction scalar divisions at the head of
// every wave, in front of the kernel-argument loads -- half a microsecond of every launch before the prologue's first request.
// Kernel arguments the prologue needs, fetched with the kernel's FIRST scalar loads: hipcc sinks an argument's s_load into the block
// that uses it, so the prologue waves paid a second and a third scalar-cache round trip (the role branch, then the site's
// pointers) before their first request to memory.
#define RWKV_ARGS_NOW(...) asm volatile("" ::RWKV_ARGS_S(__VA_ARGS__))
#define RWKV_ARGS_S(...) RWKV_ARGS_PICK(__VA_ARGS__, RWKV_A8, RWKV_A7, RWKV_A6, RWKV_A5, RWKV_A4, RWKV_A3, RWKV_A2, RWKV_A1)(__VA_ARGS__)
#define RWKV_ARGS_PICK(a, b, c, d, e, f, g, h, N, ...) N
#define RWKV_A1(a) "s"(a)
#define RWKV_A2(a, ...) "s"(a), RWKV_A1(__VA_ARGS__)
#define RWKV_A3(a, ...) "s"(a), RWKV_A2(__VA_ARGS__)
#define RWKV_A4(a, ...) "s"(a), RWKV_A3(__VA_ARGS__)
#define RWKV_A5(a, ...) "s"(a), RWKV_A4(__VA_ARGS__)
#define RWKV_A6(a, ...) "s"(a), RWKV_A5(__VA_ARGS__)
#define RWKV_A7(a, ...) "s"(a), RWKV_A6(__VA_ARGS__)
#define RWKV_A8(a, ...) "s"(a), RWKV_A7(__VA_ARGS__)
// the wave's index in its workgroup as a wave-UNIFORM value.  (threadIdx.x >> 6 is uniform in fact but "divergent" to the compiler: a role
// branch on it -- loader / prologue / consumer -- is then compiled as exec-masked straight-line code, every wave walks through every
// role's blocks, and hipcc's path-insensitive s_waitcnt insertion lets one role's pending loads tax another's: round 5 found an
// s_waitcnt vmcnt(0) in k_att's LOADER loop -- draining its DMA queue -- that guarded a register a consumer-side load had been given.)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int block_lo(int n) { return (int)(((unsigned)blockIdx.x * (unsigned)n) / gridDim.x); }
__device__ __forceinline__ int block_hi(int n) { return (int)((((unsigned)blockIdx.x + 1u) * (unsigned)n) / gridDim.x); }
constexpr unsigned VOCAB = 50277u;
constexpr float QLIM = 4194000.0f;       // |quantised activation| <= QLIM < 2^22
constexpr double QOFF = 4194304.0;       // 2^22: limbs hold q + 2^22 as an unsigned 23-bit number
constexpr float QMAGIC = 12582912.0f;    // 1.5 * 2^23: float(q + QMAGIC) has q + 2^22 in its low three bytes

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// per-token control block, lives in device memory so a captured hipGraph can be replayed for
// any token / state slot / logits row (and so the device-side argmax can feed the next step)
struct Ctl {
    unsigned long long token;
    unsigned int slot;     // state slot (PARRALEL mode: sequence index; GPT mode: 0)
    unsigned int out_row;  // logits row
    unsigned int step;     // greedy-decode step counter
    unsigned int pad;
};

// 64-lane reductions run on the DPP network (VALU-latency steps) instead of six dependent
// ds_bpermute round trips through the LDS crossbar per reduction: xor-1 / xor-2 inside quads,
// rotate by 4 and 8 inside each row of 16, then row_bcast:15 / row_bcast:31 fold the four rows into
// lane 63; readlane makes the result wave-uniform.  Lanes disabled by a row mask read 0 (`old`).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, BOUND);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, BOUND);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1, 0xf, true>(v);
    v += dpp_f64<0x4E, 0xf, true>(v);
    v += dpp_f64<0x124, 0xf, true>(v);
    v += dpp_f64<0x128, 0xf, true>(v);
    v += dpp_f64<0x142, 0xa, false>(v);
    v += dpp_f64<0x143, 0xc, false>(v);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
// max of NON-NEGATIVE values (disabled lanes contribute 0)
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f32<0xB1, 0xf, true>(v));
    v = fmaxf(v, dpp_f32<0x4E, 0xf, true>(v));
    v = fmaxf(v, dpp_f32<0x124, 0xf, true>(v));
    v = fmaxf(v, dpp_f32<0x128, 0xf, true>(v));
    v = fmaxf(v, dpp_f32<0x142, 0xa, false>(v));
    v = fmaxf(v, dpp_f32<0x143, 0xc, false>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ unsigned wave_sum_dpp(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, true);   // row_ror:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);   // row_ror:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// workgroup-wide sum of K doubles; every thread gets the totals.  `red` = NW*K doubles of LDS that no
// later reduction of the same kernel reuses before all waves have read them (callers hand out
// distinct regions of the 1 KiB scratch), so ONE barrier per reduction.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *red)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[w * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; i++) s += red[i * K + k];
        v[k] = s;
    }
}
// workgroup-wide max of K floats (same scratch discipline)
template <int K>
__device__ __forceinline__ void block_max(float (&v)[K], double *redd)
{
    float *red = reinterpret_cast<float *>(redd);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_max(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[w * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i++) s = fmaxf(s, red[i * K + k]);
        v[k] = s;
    }
}
// scratch regions (in doubles) of the RED_BYTES block: NW*K <= 24 doubles each
constexpr int RED_STATS = 0, RED_MAX = 24, RED_OFFS = 48, RED_PART = 72, RED_AUX = 96;

// ------------------------------------------------------------------------------------------
// Staged activation vectors.  Element j of a vector belongs to the 16-element piece c = j/16,
// which lane c%64 consumes at step c/64 together with weight bytes [16c, 16c+16) of a row.
// LDS image per vector: [step][limb 0..2][lane][4 dwords]; dword q of a lane holds limb b of
// elements 16c+4q .. 16c+4q+3 (byte e = element 4q+e), i.e. exactly the operand v_dot4_u32_u8
// needs against dword q of the lane's 16 weight bytes.  A lane's ds_read_b128 of one limb plane is
// lane-consecutive (conflict-free).  S*768 dwords (3 KiB per step) per vector.
template <int S> __device__ __forceinline__ constexpr int xvd() { return S * 768; }
template <int S> __device__ __forceinline__ constexpr int nquads() { return (S * 256 + NT - 1) / NT; }   // per thread

// quantise 4 consecutive elements (quad qd = j/4) with 1/scale `inv_s` and store their 3 limb dwords.
// real == false writes zero limbs: padding must contribute nothing to the integer sums (it is
// outside the row sums used for the 2^22 offset correction).
__device__ __forceinline__ void stage_quad(unsigned *xq, int qd, const float (&xr)[4], float inv_s, bool real)
{
    // One fma per element does scale, round-to-nearest-even and offset: for |x * inv_s| < 2^22 the
    // float x * inv_s + 1.5 * 2^23 has ulp 1 and bit pattern 0x4B400000 + q, i.e. its bytes 0..2 are the
    // three limbs of q + 2^22.  v_perm_b32 then transposes 4 elements x 3 bytes into 3 limb dwords.
    unsigned t[4];
#pragma unroll
    for (int e = 0; e < 4; e++) t[e] = __float_as_uint(fmaf(xr[e], inv_s, QMAGIC));
    const unsigned p01 = __builtin_amdgcn_perm(t[1], t[0], 0x05010400u);   // {t0.b0, t1.b0, t0.b1, t1.b1}
    const unsigned p23 = __builtin_amdgcn_perm(t[3], t[2], 0x05010400u);
    const unsigned h01 = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u);   // {t0.b2, t1.b2, 0, 0}
    const unsigned h23 = __builtin_amdgcn_perm(t[3], t[2], 0x0c0c0602u);
    const unsigned d0 = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    const unsigned d1 = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
    const unsigned d2 = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
    const int c = qd >> 2, q = qd & 3;
    unsigned *p = xq + ((((c >> 6) * 3) * 64 + (c & 63)) << 2) + q;
    p[0] = real ? d0 : 0u;
    p[256] = real ? d1 : 0u;
    p[512] = real ? d2 : 0u;
}
__device__ __forceinline__ float inv_scale(float amax) { return QLIM / fmaxf(amax, 1e-30f); }
__device__ __forceinline__ double scale_of(float amax) { return (double)fmaxf(amax, 1e-30f) / (double)QLIM; }
// integer sums of one row -> real value of sum_j u_j * x_j
__device__ __forceinline__ float row_value(unsigned long long T, unsigned rowsum, double scale)
{
    return (float)(scale * ((double)(long long)T - QOFF * (double)rowsum));
}

// which staged vector row r of a group multiplies with
enum { PAT_SHARED = 0,   // all rows use vector 0              (att_out, head, mm8)
       PAT_PER_ROW = 1,  // row r uses vector r                (K/V/R; the 4 quarter-rows of ffn_v)
       PAT_FFN_RK = 2 }; // rows 0..3 vector 0, row 4 vector 1 (4 ffn_k rows + 1 ffn_r row)
template <int PAT> __device__ __forceinline__ constexpr int xsel(int r)
{
    return PAT == PAT_SHARED ? 0 : (PAT == PAT_PER_ROW ? r : (r < 4 ? 0 : 1));
}
template <int PAT, int R> __device__ __forceinline__ constexpr int nvec()
{
    return PAT == PAT_SHARED ? 1 : (PAT == PAT_PER_ROW ? R : 2);
}

// ------------------------------------------------------------------------------------------
// Row-group streaming.  A group = R rows `stride` bytes apart, each `chunks` 16-byte pieces long;
// piece c belongs to lane c%64 at step c/64, so one step of one row is a 1 KiB coalesced
// non-temporal wave load.  Lanes past the end of a row (rows that are not a multiple of 1 KiB)
// re-read the row's last piece instead of branching: the staged limbs are zero there, and a
// predicated load would make hipcc drain vmcnt(0) at the branch join.
template <int R, int S>
__device__ __forceinline__ void step_load(u32x4 (&w)[R][S], int s, const uint8_t *__restrict__ base,
                                          size_t stride, int chunks, int lane, unsigned mask = 0xffffffffu)
{
    // address = wave-uniform row pointer (SGPR pair) + 32-bit per-lane byte offset (one VGPR per
    // step): global_load_dwordx4 v, v_off, s[base] -- no 64-bit per-(row,step) address registers.
    // mask == 0 turns the step into R loads of one 16-byte piece (see group_dot's refill).
    int c = lane + 64 * s;
    c = c < chunks ? c : chunks - 1;
    const unsigned off = ((unsigned)c << 4) & mask;
#pragma unroll
    for (int r = 0; r < R; r++)
        w[r][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + r * stride + off));
}
// steps [S0, S1) of a group, in consumption order (step-major)
template <int R, int S, int S0, int S1>
__device__ __forceinline__ void group_load(u32x4 (&w)[R][S], const uint8_t *__restrict__ base,
                                           size_t stride, int chunks, int lane)
{
#pragma unroll
    for (int s = S0; s < S1; s++) step_load<R, S>(w, s, base, stride, chunks, lane);
}
// How many steps of the first group are requested BEFORE the prologue runs.  A wave that asks for
// more than the memory pipe accepts stalls at issue (in-order) and its prologue waits with it; a
// wave that asks for too little leaves HBM idle while the prologue computes.  Tuned on MI355X.
#ifndef RWKV_PRE_STEPS
#define RWKV_PRE_STEPS 2
#endif
template <int S> __device__ __forceinline__ constexpr int pre_steps() { return RWKV_PRE_STEPS < S ? RWKV_PRE_STEPS : S; }

// Integer dot products of the loaded group with the staged vector(s): T[r] = sum_j u_rj * q'_j
// (q' = 23-bit unsigned limb value), every lane gets all R sums.
//  * The LDS reads of the limb pieces are software-pipelined one (step, vector) item ahead and
//    pinned with sched barriers, so at most two 12-dword pieces are live.
//  * Refill: as soon as step s of this group has been consumed its registers are re-loaded with
//    step s of the NEXT group (`next`), so the wave keeps R*S loads in flight across groups
//    instead of draining and restarting the memory pipe at every group boundary.  The refill is
//    unconditional and branch-free (a branch would make hipcc's waitcnt pass merge the two paths
//    and wait for the refill itself; two template copies in sibling branches get their common
//    code hoisted and spilled): after a wave's LAST group, `next_valid` = false degrades the refill
//    to R*S loads of one and the same 16-byte piece (one L1-resident line), which nobody waits for.
template <int R, int S, int PAT, bool REFILL = true>
__device__ __forceinline__ void group_dot(u32x4 (&w)[R][S], const unsigned *xq, int lane, unsigned long long (&T)[R],
                                          const uint8_t *__restrict__ next, size_t stride, int chunks, bool next_valid)
{
    constexpr int NV = nvec<PAT, R>();
    constexpr int NI = S * NV;
    constexpr int XVD = xvd<S>();
    const unsigned mask = next_valid ? 0xffffffffu : 0u;
    const size_t nstride = next_valid ? stride : 0;
    unsigned acc[R][3];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = 0u; }
    u32x4 x[2][3];
    const u32x4 *xb = reinterpret_cast<const u32x4 *>(xq) + lane;
#pragma unroll
    for (int b = 0; b < 3; b++) x[0][b] = xb[b * 64];
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int s = it / NV, v = it % NV;
        if (it + 1 < NI) {
            const int s1 = (it + 1) / NV, v1 = (it + 1) % NV;
            const u32x4 *p = xb + (v1 * (XVD >> 2)) + (s1 * 3) * 64;
#pragma unroll
            for (int b = 0; b < 3; b++) x[(it + 1) & 1][b] = p[b * 64];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (xsel<PAT>(r) == v) {
#pragma unroll
                for (int b = 0; b < 3; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        acc[r][b] = __builtin_amdgcn_udot4(w[r][s][q], x[it & 1][b][q], acc[r][b], false);
                // pin the partial sums here so the dot products stay between their LDS reads
                asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]));
            }
        __builtin_amdgcn_sched_barrier(0);
        if (REFILL && v == NV - 1) {
            step_load<R, S>(w, s, next, nstride, chunks, lane, mask);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // each limb total fits 32 bits (N * 255 * 255 < 2^32 for N <= 66051), so the three planes are
    // reduced as u32 and combined afterwards on wave-uniform values
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned t0 = wave_sum_dpp(acc[r][0]), t1 = wave_sum_dpp(acc[r][1]), t2 = wave_sum_dpp(acc[r][2]);
        T[r] = (unsigned long long)t0 + ((unsigned long long)t1 << 8) + ((unsigned long long)t2 << 16);
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm "sites": where a LayerNorm + token-shift mix feeds a GEMV (ln1 -> K/V/R, ln2 -> ffn k/r,
// ln_out -> head).  Every consumer workgroup needs the whole mixed vector, and computing it per
// workgroup from scratch (x, state, ~10 parameter vectors, two dependent reductions) was the largest
// loss of the first design (phase timeline: ~6 us before the first dot of a 22 us kernel).  So the
// work is split by OWNERSHIP.  With xhat = (x - mean) * rstd the staged vector of a site is
//     v_m[j] = r_m (mix_m (lnw xhat + lnb) + (1 - mix_m) prev) = C_m[j] * xhat[j] + B_m[j]
//     C_m = r_m mix_m lnw                          (static, packed at load)
//     B_m = r_m (mix_m lnb + (1 - mix_m) prev[j])  (needs only the OLD state: emitted by the row owner)
// and the offset scalar  S_m = sum_j o_m f_m[j] = rstd * PA_m - mean rstd * TC_m + PB_m  with
// PA_m = sum Co_m x,  PB_m = sum (BoL_m + BoP_m prev),  TC_m = sum Co_m  (Co = o mix lnw, ...).
// The kernel whose lanes OWN rows of x (k_first / k_attout / k_ffnv) emits B_m[j] for its rows and one
// partial tuple per workgroup {sum x, sum x^2, PA_m, PB_m; max|x|, max|B_m|}; the consumer reduces
// the tuples in ONE round (mean, rstd, S_m, and an upper bound of max|v_m| for the fixed-point scale:
// maxC_m (max|x| + |mean|) rstd + max|B_m|), then only does C*xhat + B per element.  The new state
// (ln output of this token) is committed by the row owner that overwrites x, from (mean, rstd)
// published by the consumer.  Prologue traffic drops from 48-64 to 24-32 bytes per channel.
// optional phase timeline (debug / tuning): lane 0 of every wave stamps the 100 MHz wall clock
// into tl[((block * NW) + wave) * 8 + phase].  tl == nullptr in production.
__device__ __forceinline__ void tl_stamp(unsigned long long *tl, int phase)
{
    if (tl) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = wall_clock64();
        if ((threadIdx.x & 63) == 0) tl[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + phase] = t;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NV> __host__ __device__ constexpr int site_pw() { return NV == 3 ? 16 : NV == 2 ? 12 : 8; }
struct SiteStatic {               // one site of one layer
    const float *C;               // [NV][D]  consumer side
    const float *P;               // [D][PW]  producer side: per channel, per vector m: BL, BP, Co, BoL, BoP at [m*5 + k]
    const double *TC;             // [NV]
    const float *maxC;            // [NV]
    double invD, invDm1;          // 1/D, 1/(D-1)
};
struct SiteDyn {
    float *B;                     // [NV][D]
    double *pd;                   // [n_part][8]: sum x, sum x^2, PA_0..2, PB_0..2
    float *pf;                    // [n_part][4]: max|x|, max|B_0..2|
    double *lnstat;               // [2]: mean, rstd of this site (written by consumer workgroup 0)
    int n_part;
};
template <int NV> struct SiteAcc {
    double d[8];
    float f[4];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) f[k] = 0.f;
    }
};
template <int NV> struct SitePre { f32x4 p[site_pw<NV>() / 4]; };
// row owner, before its dot product: request the producer-side constants of channel j
template <int NV>
__device__ __forceinline__ void site_prefetch(const SiteStatic &st, int j, SitePre<NV> &pre)
{
    constexpr int PW = site_pw<NV>();
#pragma unroll
    for (int k = 0; k < PW / 4; k++) pre.p[k] = reinterpret_cast<const f32x4 *>(st.P + (size_t)j * PW)[k];
}
// row owner: emit B_m[j] and accumulate this thread's share of the workgroup tuple
template <int NV>
__device__ __forceinline__ void site_emit(const SitePre<NV> &pre, const SiteDyn &dy, int D, int j, double x, double prev, SiteAcc<NV> &acc)
{
    acc.d[0] += x;
    acc.d[1] += x * x;
    acc.f[0] = fmaxf(acc.f[0], (float)fabs(x) * 1.0000002f);   // round-up guard: used as an upper bound
#pragma unroll
    for (int m = 0; m < NV; m++) {
        const float bl = pre.p[(m * 5 + 0) >> 2][(m * 5 + 0) & 3], bp = pre.p[(m * 5 + 1) >> 2][(m * 5 + 1) & 3];
        const float co = pre.p[(m * 5 + 2) >> 2][(m * 5 + 2) & 3];
        const float bol = pre.p[(m * 5 + 3) >> 2][(m * 5 + 3) & 3], bop = pre.p[(m * 5 + 4) >> 2][(m * 5 + 4) & 3];
        const float b = (float)((double)bl + (double)bp * prev);
        dy.B[(size_t)m * D + j] = b;
        acc.d[2 + m] += (double)co * x;
        acc.d[5 + m] += (double)bol + (double)bop * prev;
        acc.f[1 + m] = fmaxf(acc.f[1 + m], fabsf(b));
    }
}
// all threads: fold the accumulators into this workgroup's tuple.  LIVE = lanes per wave that hold
// one (the row owners are lanes 0..R-1 of every wave; 64 = every lane).  scratch: NW*LIVE*12 words of
// LDS (or the 640-byte reduction block for LIVE == 64) that no wave still reads.
template <int NV, int LIVE>
__device__ __forceinline__ void site_publish(SiteAcc<NV> &acc, const SiteDyn &dy, void *scratch)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NE = LIVE == 64 ? NW : NW * LIVE;   // entries to add up
    double *red = reinterpret_cast<double *>(scratch);
    float *redf = reinterpret_cast<float *>(red + NE * 8);
    if (LIVE == 64) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < 2 || (k - 2) % 3 < NV) acc.d[k] = wave_sum(acc.d[k]);
#pragma unroll
        for (int k = 0; k < 1 + NV; k++) acc.f[k] = wave_max(acc.f[k]);
    }
    if (lane < (LIVE == 64 ? 1 : LIVE)) {
        const int e = LIVE == 64 ? w : w * LIVE + lane;
#pragma unroll
        for (int k = 0; k < 8; k++) red[e * 8 + k] = acc.d[k];
#pragma unroll
        for (int k = 0; k < 4; k++) redf[e * 4 + k] = acc.f[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
        for (int i = 0; i < NE; i++) t += red[i * 8 + threadIdx.x];
        dy.pd[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
    } else if (threadIdx.x < 12) {
        float t = 0.f;
        for (int i = 0; i < NE; i++) t = fmaxf(t, redf[i * 4 + (threadIdx.x - 8)]);
        dy.pf[(size_t)blockIdx.x * 4 + (threadIdx.x - 8)] = t;
    }
}
template <int NV> struct SiteRed {
    double mean, rstd;
    double S[NV];
    float amax[NV];
};
// consumer, step 1 (first loads of the kernel): this thread's partial tuple
struct SiteTuple { f64x2 d[4]; f32x4 f; };
__device__ __forceinline__ void site_tuple_load(const SiteDyn &dy, SiteTuple &t)
{
    const int i = (int)threadIdx.x < dy.n_part ? threadIdx.x : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) t.d[k] = reinterpret_cast<const f64x2 *>(dy.pd)[i * 4 + k];
    t.f = reinterpret_cast<const f32x4 *>(dy.pf)[i];
}
// consumer, step 2: one reduction round -> LayerNorm statistics, offset scalars, scale bounds
// NWR = number of waves taking part (waves 0..NWR-1).  spin != nullptr: the waves meet on an LDS
// counter (zeroed by the caller behind an earlier barrier) instead of the workgroup barrier, so the
// remaining waves of the workgroup need not take part.
template <int NV, int NWR = NW>
__device__ __forceinline__ void site_reduce(const SiteStatic &st, const SiteDyn &dy, const SiteTuple &t, int D, double *red, SiteRed<NV> &out,
                                            const double (&tc)[NV], const float (&mc)[NV], unsigned long long *tl = nullptr, unsigned *spin = nullptr)
{
    float *redf = reinterpret_cast<float *>(red + NW * 8);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool real = (int)threadIdx.x < dy.n_part;
    double d[8];
    float f[4];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = real ? t.d[k >> 1][k & 1] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = real ? t.f[k] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < 2 || (k - 2) % 3 < NV) d[k] = wave_sum(d[k]);
#pragma unroll
    for (int k = 0; k < 1 + NV; k++) f[k] = wave_max(f[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) red[w * 8 + k] = d[k];
#pragma unroll
        for (int k = 0; k < 4; k++) redf[w * 4 + k] = f[k];
    }
    tl_stamp(tl, 3);
    if (spin) {
        if (lane == 0) __hip_atomic_fetch_add(spin, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(spin, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)NWR) __builtin_amdgcn_s_sleep(1);
    } else {
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < 2 || (k - 2) % 3 < NV) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NWR; i++) s += red[i * 8 + k];
            d[k] = s;
        }
    }
#pragma unroll
    for (int k = 0; k < 1 + NV; k++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NWR; i++) s = fmaxf(s, redf[i * 4 + k]);
        f[k] = s;
    }
    out.mean = d[0] * st.invD;
    const double var = (d[1] - d[0] * out.mean) * st.invDm1;   // reference: (D-1), no epsilon (rwkv.cu:43-44,53)
    out.rstd = rsqrt(var);
    const double mrs = out.mean * out.rstd;
#pragma unroll
    for (int m = 0; m < NV; m++) {
        out.S[m] = out.rstd * d[2 + m] - mrs * tc[m] + d[5 + m];
        out.amax[m] = (mc[m] * (float)(((double)f[0] + fabs(out.mean)) * out.rstd) + f[1 + m]) * 1.0001f;
    }
}
__device__ __forceinline__ void load_quad_f64(const double *p, int qd, double (&out)[4])
{
    const f64x2 a = reinterpret_cast<const f64x2 *>(p)[qd * 2], b = reinterpret_cast<const f64x2 *>(p)[qd * 2 + 1];
    out[0] = a[0]; out[1] = a[1]; out[2] = b[0]; out[3] = b[1];
}
// consumer, step 3: v_m = C_m * xhat + B_m for this thread's quads, quantised and staged
template <int NV, int NQ, int S, int NTP = NT>
__device__ __forceinline__ void site_stage(const double (&xl)[NQ][4], const f32x4 (&Cq)[NQ][NV], const f32x4 (&Bq)[NQ][NV],
                                           const SiteRed<NV> &sr, unsigned *xq, int nqd)
{
    constexpr int XVD = xvd<S>();
    const float rstdf = (float)sr.rstd;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NTP;
        if (qd < S * 256) {
            float xh[4];   // the subtraction stays in f64 (|mean| may dwarf the deviation); the rest is f32
#pragma unroll
            for (int e = 0; e < 4; e++) xh[e] = (float)(xl[i][e] - sr.mean) * rstdf;
#pragma unroll
            for (int m = 0; m < NV; m++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(Cq[i][m][e], xh[e], Bq[i][m][e]);
                stage_quad(xq + m * XVD, qd, v, inv_scale(sr.amax[m]), qd < nqd);
            }
        }
    }
}
// ------------------------------------------------------------------------------------------
// Kernel prologues.  What the first microseconds of a kernel look like decides a third of its run
// time, and three measured properties of the CU's vector-memory path shape them:
//  * data returns IN ORDER per CU (no hit-under-miss): a prologue load queued behind weight loads
//    arrives after them, however warm its line is in L2;
//  * the path holds only ~16 KB of requests: a wave that asks for more blocks at issue until HBM has
//    delivered -- and a blocked wave cannot run the prologue or reach a barrier;
//  * every workgroup reads the same ~100 KB of prologue inputs, 3.6 MB per XCD through one L2.
// SPLIT prologues therefore give the waves roles until the vector is staged: waves 0..NW/2-1 request
// the prologue inputs of the WHOLE vector, an order barrier lets those requests enter the pipe first,
// then waves NW/2.. ("loaders") request their entire first row group -- they block at issue, which
// is the point: the weight stream starts in the second microsecond and runs under the prologue --
// while the prologue waves reduce (meeting on an LDS counter, not the workgroup barrier), quantise
// and stage, publish the scalars through LDS, and join the loaders at the "staged" barrier.
// which kernels use the role split (bit 0 k_att, 1 k_attout, 2 k_ffn_rk, 3 k_ffnv, 4 k_head): measured per kernel
#ifndef RWKV_SPLIT
#define RWKV_SPLIT 23
#endif
constexpr int RED_BC = 96;     // doubles: scalars published by the prologue waves (8 floats) + spin counter

// Row groups of a workgroup are handed out through an LDS counter (DYN): the waves that ran the
// prologue start their first group late, the others take more of the remaining groups.  The counter
// lives next to the published scalars and is set by thread 0 before the prologue's barriers.
#ifndef RWKV_HEAD_R
#define RWKV_HEAD_R 2      // rows per group in k_head (measured: 2 -> 35.8, 3 -> 36.2, 4 -> 36.8, 5 -> 38.0 us at 7B)
#endif
#ifndef RWKV_HEAD_RR
#define RWKV_HEAD_RR 4     // the same in the ring variant (a group = one LDS slot: larger groups, fewer hand-offs)
#endif
// Row-group buffers per wave: 1, or 2 (a wave holds TWO groups of weight registers, A and B, and alternates between
// them: R*S*2 loads requested ahead).  Measured on MI355X (profiles/r02/decode_variants.txt) two buffers LOSE 8 % at 7B
// (485 vs 528 tokens/s): the CU's memory pipe accepts only ~16 KB of requests, so a loader wave that asks for two groups
// sits blocked at issue twice as long and the "staged" barrier, which it must reach too, moves from 5 to 10 us, while the
// stream was never starved with one (the loaders' 80 KB per CU last until the vector is staged).  Kept as a knob.
#ifndef RWKV_NBUF
#define RWKV_NBUF 1
#endif
constexpr int NBUF = RWKV_NBUF;          // default; a kernel whose two buffers would not fit 256 registers (R*S*8 of them) takes 1
constexpr int nb_att(int) { return NBUF; }                        // 3 rows:  254 registers at S = 5
constexpr int nb_attout(int) { return NBUF; }
constexpr int nb_frk(int S) { return S <= 4 ? NBUF : 1; }         // 5 rows:  245 at S = 4, spills at S = 5
constexpr int nb_fv(int S) { return S <= 4 ? NBUF : 1; }          // 4 rows:  229 at S = 4, spills at S = 5
constexpr int nb_head(int) { return NBUF; }
__device__ __forceinline__ unsigned *group_counter(double *red) { return reinterpret_cast<unsigned *>(red + RED_BC) + 9; }
__device__ __forceinline__ int next_group(unsigned *ctr)
{
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane((int)v);
}

// The streaming loop of every decode kernel.  A wave owns the groups it draws from the workgroup's LDS counter; group
// indices drawn later are larger, so the OLDEST outstanding group is tested and a miss ends the loop.  Per group:
// pre(g) requests the epilogue's inputs (before the dot issues the refills: a load placed after them would, by in-order
// vmcnt, wait for the whole next group to land), group_dot consumes the group's registers and refills them step by step
// with the group drawn now, epi(g, T, in) finishes the rows.  base(g) = weight address of a valid group.
template <int R, int S, int PAT, int NB, class Base, class Pre, class Epi>
__device__ __forceinline__ void stream_groups(u32x4 (&wA)[R][S], u32x4 (&wB)[R][S], int gA, int gB, int g0, int g1, unsigned *gctr,
                                              const unsigned *xq, int lane, size_t stride, int chunks, Base base, Pre pre, Epi epi)
{
    for (;;) {
        if (!(gA < g1)) break;
        {
            const int gn = next_group(gctr);
            const bool nv = gn < g1;
            const auto in = pre(gA);
            unsigned long long T[R];
            group_dot<R, S, PAT>(wA, xq, lane, T, base(nv ? gn : g0), stride, chunks, nv);
            epi(gA, T, in);
            gA = gn;
        }
        if (NB == 2) {
            if (!(gB < g1)) break;
            const int gn = next_group(gctr);
            const bool nv = gn < g1;
            const auto in = pre(gB);
            unsigned long long T[R];
            group_dot<R, S, PAT>(wB, xq, lane, T, base(nv ? gn : g0), stride, chunks, nv);
            epi(gB, T, in);
            gB = gn;
        }
    }
}
// first groups of a wave (A: g0 + wave, B: g0 + NW + wave) and the counter's start; set by thread 0 before the prologue's barriers
template <int NB>
__device__ __forceinline__ void first_groups(int g0, int wave, unsigned *gctr, int &gA, int &gB)
{
    gA = g0 + wave;
    gB = NB == 2 ? g0 + NW + wave : 0x7fffffff;
    if (threadIdx.x == 0) *gctr = (unsigned)(g0 + NB * NW);
}
// workgroup-wide sum of one double and max of one non-negative float in ONE barrier (the kernels' closing reduction)
__device__ __forceinline__ void block_sum_max(double &sv, float &mv, double *red)
{
    float *redf = reinterpret_cast<float *>(red + NW);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    sv = wave_sum(sv);
    mv = wave_max(mv);
    if (lane == 0) { red[w] = sv; redf[w] = mv; }
    __syncthreads();
    double s = 0.0; float m = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) { s += red[i]; m = fmaxf(m, redf[i]); }
    sv = s; mv = m;
}

// LayerNorm-site consumer (k_att, k_ffn_rk, k_head): on return the NV vectors are staged in xq,
// sr.S / sr.amax are valid in every wave and w holds (requests for) the wave's first row group.
template <int NV, int R, int S, bool SPLIT, int NB>
__device__ __forceinline__ void site_open(const SiteStatic &st, const SiteDyn &dy, const double *x, int D, double *red, unsigned *xq,
                                          u32x4 (&w)[R][S], u32x4 (&w2)[R][S], const uint8_t *wb, const uint8_t *wb2, size_t stride,
                                          SiteRed<NV> &sr, bool publish_stats, unsigned long long *tl)
{
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4, nqd = D >> 2;
    constexpr int NTP = SPLIT ? NT / 2 : NT, NWP = NTP / 64, NQP = (S * 256 + NTP - 1) / NTP;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    if (SPLIT && wave >= NWP) {
        __syncthreads();   // order: the prologue waves' requests are in the memory pipe
        group_load<R, S, 0, S>(w, wb, stride, chunks, lane);
        if (NB == 2) group_load<R, S, 0, S>(w2, wb2, stride, chunks, lane);
        tl_stamp(tl, 2);
        __syncthreads();   // staged
    } else {
        // the site's static scalars are read here, not where they are used: behind the reduction's barrier the
        // compiler may not hoist them, and a cold scalar load there sits on the prologue's critical path
        double tc[NV];
        float mc[NV];
#pragma unroll
        for (int m = 0; m < NV; m++) { tc[m] = st.TC[m]; mc[m] = st.maxC[m]; }
        SiteTuple tup;
        site_tuple_load(dy, tup);
        double xl[NQP][4];
        f32x4 Cq[NQP][NV], Bq[NQP][NV];
#pragma unroll
        for (int i = 0; i < NQP; i++) {
            const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
            load_quad_f64(x, qc, xl[i]);
#pragma unroll
            for (int m = 0; m < NV; m++) {
                Cq[i][m] = reinterpret_cast<const f32x4 *>(st.C + (size_t)m * D)[qc];
                Bq[i][m] = reinterpret_cast<const f32x4 *>(dy.B + (size_t)m * D)[qc];
            }
        }
        tl_stamp(tl, 1);
        if (SPLIT) {
            if (threadIdx.x == 0) *spin = 0u;
            __syncthreads();   // order (see the loader role)
            site_reduce<NV, NWP>(st, dy, tup, D, red, sr, tc, mc, tl, spin);
        } else {
            // a small pre-issue of the weight stream; more would block this wave's prologue
            group_load<R, S, 0, pre_steps<S>()>(w, wb, stride, chunks, lane);
            tl_stamp(tl, 2);
            site_reduce<NV, NWP>(st, dy, tup, D, red, sr, tc, mc, tl);
        }
        if (publish_stats && blockIdx.x == 0 && threadIdx.x == 0) { dy.lnstat[0] = sr.mean; dy.lnstat[1] = sr.rstd; }
        tl_stamp(tl, 4);
        site_stage<NV, NQP, S, NTP>(xl, Cq, Bq, sr, xq, nqd);
        if (SPLIT && threadIdx.x == 0) {
#pragma unroll
            for (int m = 0; m < NV; m++) { bc[m] = (float)sr.S[m]; bc[4 + m] = sr.amax[m]; }
        }
        __syncthreads();   // staged
        group_load<R, S, SPLIT ? 0 : pre_steps<S>(), S>(w, wb, stride, chunks, lane);
        if (NB == 2) group_load<R, S, 0, S>(w2, wb2, stride, chunks, lane);
    }
    if (SPLIT) {
#pragma unroll
        for (int m = 0; m < NV; m++) { sr.S[m] = (double)bc[m]; sr.amax[m] = bc[4 + m]; }
    }
    tl_stamp(tl, 5);
}

// Plain-vector consumer (k_attout: NVEC = 1; k_ffnv: the four quarter vectors of the hidden vector).
// The producer left the pre-scaled vector and per-workgroup partials of the offset sum and of max|.|.
template <int NVEC, int R, int S, bool SPLIT, int NB>
__device__ __forceinline__ void vec_open(const float *vec, const double *partS, const float *partM, int n_part, int D, double *red,
                                         unsigned *xq, u32x4 (&w)[R][S], u32x4 (&w2)[R][S], const uint8_t *wb, const uint8_t *wb2, size_t stride,
                                         float &Sf, float &amax, unsigned long long *tl)
{
    constexpr int XVD = xvd<S>();
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4, nqd = D >> 2;
    constexpr int NTP = SPLIT ? NT / 2 : NT, NWP = NTP / 64, NQP = (S * 256 + NTP - 1) / NTP;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    if (SPLIT && wave >= NWP) {
        __syncthreads();   // order
        group_load<R, S, 0, S>(w, wb, stride, chunks, lane);
        if (NB == 2) group_load<R, S, 0, S>(w2, wb2, stride, chunks, lane);
        tl_stamp(tl, 2);
        __syncthreads();   // staged
    } else {
        double ps = partS[(int)threadIdx.x < n_part ? threadIdx.x : 0];
        float pm = partM[(int)threadIdx.x < n_part ? threadIdx.x : 0];
        float vl[NVEC][NQP][4];
#pragma unroll
        for (int q = 0; q < NVEC; q++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
                const f32x4 t = reinterpret_cast<const f32x4 *>(vec + (size_t)q * D)[qc];
                vl[q][i][0] = t[0]; vl[q][i][1] = t[1]; vl[q][i][2] = t[2]; vl[q][i][3] = t[3];
            }
        tl_stamp(tl, 1);
        if (SPLIT) {
            if (threadIdx.x == 0) *spin = 0u;
            __syncthreads();   // order
        } else {
            group_load<R, S, 0, pre_steps<S>()>(w, wb, stride, chunks, lane);
            tl_stamp(tl, 2);
        }
        if ((int)threadIdx.x >= n_part) { ps = 0.0; pm = 0.f; }
        {   // scale and offset from the producer's per-workgroup partials: one reduction round
            float *redf = reinterpret_cast<float *>(red + RED_MAX);
            const double ws = wave_sum(ps);
            const float wm = wave_max(pm);
            if (lane == 0) { red[RED_OFFS + wave] = ws; redf[wave] = wm; }
            tl_stamp(tl, 3);
            if (SPLIT) {
                if (lane == 0) __hip_atomic_fetch_add(spin, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(spin, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)NWP) __builtin_amdgcn_s_sleep(1);
            } else {
                __syncthreads();
            }
            double ts = 0.0; float tm = 0.f;
#pragma unroll
            for (int i = 0; i < NWP; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
            Sf = (float)ts; amax = tm;
        }
        tl_stamp(tl, 4);
#pragma unroll
        for (int q = 0; q < NVEC; q++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP;
                if (qd < S * 256) stage_quad(xq + q * XVD, qd, vl[q][i], inv_scale(amax), qd < nqd);
            }
        if (SPLIT && threadIdx.x == 0) { bc[0] = Sf; bc[4] = amax; }
        __syncthreads();   // staged
        group_load<R, S, SPLIT ? 0 : pre_steps<S>(), S>(w, wb, stride, chunks, lane);
        if (NB == 2) group_load<R, S, 0, S>(w2, wb2, stride, chunks, lane);
    }
    if (SPLIT) { Sf = bc[0]; amax = bc[4]; }
    tl_stamp(tl, 5);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA streaming ("ring" kernels: template parameter RING of the five decode kernels, RWKV_RING bit per kernel class).
// Loads into registers top out at ~10 B/clk per CU on this chip whatever is kept in flight;
// global_load_lds_dwordx4 -- the CU's DMA path from memory into LDS -- does not return through the vector register file.
// The workgroup's row groups travel through a ring of `ns` LDS slots (one group of R rows x S KiB each):
//   * the LAST wave of the workgroup is the loader: it issues the DMA of group k into slot k % ns -- R*S wave instructions
//     of 1 KiB -- keeps up to DEPTH groups (<= 63 instructions, the vmcnt range) in flight and publishes a group
//     (ready[slot] = k + 1) once its vmcnt says the group has landed (data returns in order);
//   * the other NW-1 waves consume: group k belongs to wave k % (NW-1); it waits for ready[slot], copies the slot into
//     registers (R*S ds_read_b128), hands the slot back (freeq[slot] = k + 1) and runs the same group_dot + epilogue as
//     the register kernels;
//   * the loader starts at the kernel's first instruction: the ring (>= 90 KiB) holds what HBM delivers during the
//     prologue, which waves 0..3 run as in the SPLIT kernels.  No workgroup barrier sits between the order barrier and
//     the closing reduction -- the roles meet on LDS counters only, and every wait loop is bounded.
// The DMA is inline asm (hipcc would otherwise count it in vmcnt and drain it before every LDS access of the loader);
// m0 carries the wave-uniform LDS destination and is restored for the compiler.
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ void dma_piece(const uint8_t *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// the same for data that OTHER workgroups of the XCD read as well (the chunk path's activation image): no "nt", it should stay in L2
__device__ __forceinline__ void dma_piece_shared(const uint8_t *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// One row of S KiB as S DMA instructions that share ONE address register pair and one M0 value: the instruction offset advances
// the global address AND the LDS address (tools/dmabench.hip, modes 2 / 6: every word verified).  A one-wave DMA stream is bounded
// by the loader wave's instruction count -- at 7 instructions per piece it stood at 5.7 TB/s chip-wide, at 2 per piece 7.0 TB/s.
template <int S> __device__ __forceinline__ void dma_unit(const uint8_t *src, unsigned lds_dst);
#define RWKV_DMA_UNIT(S_, BODY)                                                                                                        \
    template <> __device__ __forceinline__ void dma_unit<S_>(const uint8_t *src, unsigned lds_dst)                                     \
    {                                                                                                                                  \
        unsigned keep;                                                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" BODY "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory"); \
    }
#define RWKV_DMA_L(OFF) "global_load_lds_dwordx4 %1, off offset:" #OFF " nt\n\t"
RWKV_DMA_UNIT(1, RWKV_DMA_L(0))
RWKV_DMA_UNIT(2, RWKV_DMA_L(0) RWKV_DMA_L(1024))
RWKV_DMA_UNIT(3, RWKV_DMA_L(0) RWKV_DMA_L(1024) RWKV_DMA_L(2048))
RWKV_DMA_UNIT(4, RWKV_DMA_L(0) RWKV_DMA_L(1024) RWKV_DMA_L(2048) RWKV_DMA_L(3072))
template <> __device__ __forceinline__ void dma_unit<5>(const uint8_t *src, unsigned lds_dst)      // the offset field ends at 4095
{
    dma_unit<4>(src, lds_dst);
    dma_unit<1>(src + 4096, lds_dst + 4096);
}
#undef RWKV_DMA_L
#undef RWKV_DMA_UNIT
// First statement of every loader wave.  hipcc does not see the loader's DMA (inline asm), but it does remember every vector memory
// operation the wave issued BEFORE the role branch -- the debug timeline's store, a kernel-entry load -- and guards the registers those
// were given with s_waitcnt vmcnt(0) wherever the loader's code first reuses them: by the luck of register allocation that was inside
// the loader's loop in one round-5 build of k_att (12.3 -> 16.8 us: a drain of the DMA queue per group).  A wait the compiler DOES see
// (the builtin, not asm), placed where nothing of the DMA is in flight yet, settles its books: nothing is pending behind it.
__device__ __forceinline__ void loader_clean_slate() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0) expcnt(7) lgkmcnt(15)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
constexpr int GLDS_SPIN = 1 << 18;    // bound of every wait loop (a lost hand-off must end the kernel, not hang the GPU)
// A wait that ran into its bound is RECORDED in a wave-uniform REGISTER (`fail`: no memory operation on any path that rejoins the
// streaming loops -- an error word stored from inside them, global or LDS, cost 3.5 % of the decode rate: hipcc's waitcnt insertion
// is path-insensitive, and the loader wave's instruction count is the stream's ceiling) and REPORTED once per wave, behind the
// kernel's closing barrier (ring_report), to the context's error word (mapped host memory; engine.hip device_check reads it after
// the stream synchronisation and fails the call with RWKV_E_DEVICE): the kernel still ends, but nobody is handed its results.
// codes: 1 loader found no room, 2 a group never landed, 3 the prologue never staged (or the carried rows were never verified),
// 4 the loader's own DMA never completed.  
__device__ __forceinline__ void wait_count(const unsigned *p, unsigned least, unsigned &fail)
{
    bool ok = false;
    for (int it = 0; it < GLDS_SPIN; it++) {
        if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= least) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    fail = ok ? fail : 3u;
}
__device__ __forceinline__ void ring_report(unsigned fail, unsigned *herr)
{
#ifndef RWKV_NO_HERR      // (A/B knob: what recording the failures costs; without the report the compiler drops every `fail` update)
    if (fail != 0u && herr != nullptr && (threadIdx.x & 63) == 0) __hip_atomic_store(herr, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
constexpr int GLDS_FQ = 32;             // groups the ring bookkeeping can hold
// DMA pieces (1 KiB) the loader keeps in flight.  By Little's law the queueing delay of EVERY access of the CU is in-flight bytes /
// stream rate: 63 KiB at 23 KB/us = 2.7 us -- paid by the prologue's loads and, after the last issue, by the kernel's tail --
// while the stream itself is at full rate from ~24 KiB up (tools/dmabench.hip: depth 63 / 32 / 16 -> 7.3 / 7.3 / 6.0 TB/s).
// Round 6, with the matrix-core consumers (profiles/r06/depth_ab.txt): 24 .. 30 are a plateau 0.5 % above 32 at 7B, 20 falls off a cliff (-7.5 %);
// 28 with 20 in front of the barrier: 7B +0.55 %, 14B +0.8 %, 1B5 / 3B +0.1-0.2 %.
#ifndef RWKV_RING_DEPTH
#define RWKV_RING_DEPTH 28
#endif
// The weight stream starts BEFORE the workgroup's order barrier: the loader issues its first RWKV_RING_PRE rows at once, thinly
// (RWKV_RING_PRE_DEPTH pieces in flight: the prologue's loads, issued ~1.4 us into the kernel, queue behind at most that), and
// only then joins the barrier that lets the prologue's requests into the pipe ahead of the deep stream.  The cold start of a
// kernel's stream (first translations, empty memory queues) then overlaps the prologue waves' own start-up.
#ifndef RWKV_RING_PRE
#define RWKV_RING_PRE 8
#endif
// TEST build only (tests/test_engine_gpu.py builds a variant with -DRWKV_TEST_DROP_GROUP=1): the loader "loses" every workgroup's last
// group, so the consumers' bounded wait must give up and the call must fail with RWKV_E_DEVICE instead of returning garbage
#ifndef RWKV_TEST_DROP_GROUP
#define RWKV_TEST_DROP_GROUP 0
#endif
#ifndef RWKV_LOADER_PRIO
#define RWKV_LOADER_PRIO 0        // s_setprio of the loader wave (0..3)
#endif
#ifndef RWKV_RING_PRE_DEPTH
#define RWKV_RING_PRE_DEPTH 20
#endif
struct GldsCtl {            // LDS control block of the ring
    unsigned staged;        // prologue waves that have staged their part of the vector
    unsigned landed;        // ring units (rows of S KiB) whose DMA has completed: loader -> consumers, monotonic
    unsigned pad[2];
    unsigned freeq[GLDS_FQ];   // freeq[k % FQ] = k + 1: group k has been copied out of the ring
    unsigned gend[GLDS_FQ];    // the loader's own: end unit of group k
};
static_assert(sizeof(GldsCtl) % 16 == 0, "the ring behind the control block stays 16-byte aligned");
constexpr int GLDS_CTL_ZERO = (int)(sizeof(GldsCtl) / 4);
constexpr int NC = NW - 1;              // consumer waves of a ring kernel
// LDS of a row-form kernel in ring form: [reduction scratch RED_BYTES][staged vectors: nv x S x 3 KiB][GldsCtl][ring: ns units of S KiB]

// The loader wave.  The ring is made of `nu` UNITS of one row (S KiB) each; a group of R rows takes the next R units (wrapping),
// so slots of every group size share one ring and the LDS is used to the last 4 KiB.  Round 2's first loader handed out whole
// group slots, counted its groups with blocking s_waitcnt and DRAINED (vmcnt(0)) whenever the consumers were behind -- which in
// the ring-full regime (every kernel here, once the prologue is over) serialised issue -> land -> announce per group: 17 KB/us per
// CU.  This one never blocks on a count it could read instead:
//   * the wave's own vmcnt is READ (s_getreg IB_STS), so "everything but the last vm pieces has landed" is announced as a
//     monotonic unit count while the wave waits for room -- the consumer that will free the wanted units may be waiting for exactly that;
//   * consumers free groups out of order (freeq flags); the tail moves over every leading free group with ONE LDS round trip
//     (lane i looks at group tail + i);
//   * a row is one asm statement (dma_unit); all state is wave-uniform.  The loader wave's instruction count IS the stream's
//     ceiling: tools/dmabench.hip, dmabench2.hip (this protocol with the real group_dot: 6.8 TB/s chip-wide, 26 KB/us per CU).
template <int S> struct RingLoader {
    GldsCtl *mc;
    unsigned ring, nu;       // LDS byte address of unit 0, units
    unsigned off[S];         // lane's byte offset in a row, per step (rows that are not whole KiB: the last pieces are clamped)
    bool whole;              // rows are whole KiB: a unit is one dma_unit
    unsigned issued = 0, pub = 0;       // units issued / announced as landed
    unsigned k = 0, tail = 0;           // groups issued / groups known to be copied out
    unsigned tailu = 0;                 // first unit still in use
    unsigned pos = 0;                   // ring position of the next unit
    int lane;
    bool dead = false;
    unsigned fail = 0;                  // see wait_count

    __device__ __forceinline__ RingLoader(GldsCtl *mc_, unsigned ring_, int nu_, int chunks, int lane_)
        : mc(mc_), ring((unsigned)__builtin_amdgcn_readfirstlane((int)ring_)), nu((unsigned)nu_), lane(lane_)
    {
        asm volatile("" : "+s"(ring));      // an opaque SGPR value (else the generic -> LDS address conversion is redone at every use)
        whole = chunks == 64 * S;
#pragma unroll
        for (int s = 0; s < S; s++) {
            int c = lane + 64 * s;
            c = c < chunks ? c : chunks - 1;
            off[s] = (unsigned)c << 4;
        }
    }
    __device__ __forceinline__ void publish(unsigned units)
    {
        if ((int)(units - pub) > 0) {
            pub = units;
            if (lane == 0) __hip_atomic_store(&mc->landed, units, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // DMA instructions of this wave that have not completed: HW_REG_IB_STS holds VM_CNT in bits 3:0 and its two high bits in 23:22.
    // Loads complete in order and this wave issues no other vector memory instruction.
    __device__ __forceinline__ unsigned in_flight() const
    {
        const unsigned v = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 7);
        return (v & 0xfu) | ((v >> 18) & 0x30u);
    }
    __device__ __forceinline__ void poll_landed()
    {
        const unsigned tot = issued * (unsigned)S, f = in_flight();
        publish(tot > f ? (tot - f) / (unsigned)S : 0u);        // (saturating: never announce more than was issued, whatever vmcnt holds)
    }
    __device__ __forceinline__ void advance_tail()
    {
        const unsigned g = tail + (unsigned)(lane & 31);
        const unsigned f = __hip_atomic_load(&mc->freeq[g % GLDS_FQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned e = __hip_atomic_load(&mc->gend[g % GLDS_FQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");      // compiler-level ordering is all LDS needs: a consumer's reads precede its flag in the LDS queue
        const bool ok = lane < 32 && (g - tail) < (k - tail) && f == g + 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        const unsigned n = (unsigned)__builtin_ctzll(~m);          // leading groups that are free (<= 32: bits 32.. are never set)
        if (n) {
            tailu = (unsigned)__builtin_amdgcn_readlane((int)e, (int)(n - 1u));
            tail += n;
        }
    }
    template <int R> __device__ __forceinline__ bool room() const { return !(issued + R - tailu > nu || k - tail >= (unsigned)GLDS_FQ); }
    // R rows starting at `src` (this lane's first piece of row 0; rows `stride` bytes apart) -> the next R units
    template <int R> __device__ __forceinline__ void group(const uint8_t *src, size_t stride, bool thin = false)
    {
        for (int it = 0; !room<R>() && !dead; it++) {
            advance_tail();
            if (room<R>()) break;
            poll_landed();
            if (it >= GLDS_SPIN) { dead = true; fail = 1u; }      // a lost hand-off must end the kernel, not hang the GPU -- and must be reported
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) mc->gend[k % GLDS_FQ] = issued + R;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (thin) wait_vm<(RWKV_RING_PRE_DEPTH > S ? RWKV_RING_PRE_DEPTH : S) - S>();
            else wait_vm<(RWKV_RING_DEPTH < 63 ? RWKV_RING_DEPTH : 63) - S>();      // 63: the counter has 6 bits
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + pos * (unsigned)(S * 1024)));
            if (whole) dma_unit<S>(src + r * stride, dst);
            else {
#pragma unroll
                for (int s = 0; s < S; s++) dma_piece(src + r * stride + (off[s] - off[0]), dst + s * 1024);
            }
            pos = pos + 1 == nu ? 0u : pos + 1;
        }
        issued += R;
        k++;
        poll_landed();
    }
    __device__ __forceinline__ void finish()
    {
        int it = 0;
        for (; it < GLDS_SPIN && in_flight() != 0u; it++) { poll_landed(); __builtin_amdgcn_s_sleep(1); }
        fail = it >= GLDS_SPIN ? 4u : fail;
        wait_vm<0>();
        publish(issued);
    }
};
// the loader wave of a launch kernel: groups [g0, g1) of the workgroup in order; base(g) = address of group g's first row
template <int R, int S, class Base>
__device__ __forceinline__ unsigned glds_loader(Base base, int g0, int g1, size_t stride, int chunks, int nu, unsigned ring, GldsCtl *ctl, int lane)
{
    loader_clean_slate();
#if RWKV_LOADER_PRIO
    __builtin_amdgcn_s_setprio(RWKV_LOADER_PRIO);     // the loader's instruction issue IS the stream's ceiling: let it win the SIMD's arbitration
#endif
    // the control block is this wave's to zero: nobody else touches it before the order barrier
    for (int i = lane; i < GLDS_CTL_ZERO; i += 64) reinterpret_cast<unsigned *>(ctl)[i] = 0u;
    RingLoader<S> ld(ctl, ring, nu, chunks, lane);
    int g = g0;
    const int pre = RWKV_RING_PRE < nu - R ? RWKV_RING_PRE : nu - R;      // never wait for room before the barrier: the consumers are behind it
    for (; g < g1 && (int)ld.issued < pre; g++) ld.template group<R>(base(g) + ld.off[0], stride, true);
    __syncthreads();   // order: the control block is zero, the prologue's requests are in the pipe
    // (TEST build -DRWKV_TEST_DROP_GROUP=1: the loader "loses" the workgroup's last group)
    for (; g < g1 - (RWKV_TEST_DROP_GROUP ? 1 : 0); g++) ld.template group<R>(base(g) + ld.off[0], stride);
    ld.finish();
    return ld.fail;
}
// consumer side of one group (kl = its index in the workgroup): wait, copy its R units into registers, hand them back
template <int R, int S>
__device__ __forceinline__ void glds_take(u32x4 (&w)[R][S], int kl, int nu, const unsigned char *ring, GldsCtl *ctl, int lane, unsigned &fail)
{
    const unsigned uend = (unsigned)(kl + 1) * R;
    bool ok = false;
    for (int it = 0; it < GLDS_SPIN; it++) {
        if ((int)(__hip_atomic_load(&ctl->landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - uend) >= 0) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    fail = ok ? fail : 2u;
    unsigned p0 = (uend - R) % (unsigned)nu;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(ring + (size_t)p0 * (S * 1024)) + lane;
#pragma unroll
        for (int s = 0; s < S; s++) w[r][s] = p[s * 64];
        p0 = p0 + 1 == (unsigned)nu ? 0u : p0 + 1;
    }
    if (lane == 0) __hip_atomic_store(&ctl->freeq[kl % GLDS_FQ], (unsigned)kl + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// EARLY TAKE (round 4).  The ring holds what HBM delivers while waves 0..3 run the prologue; when the prologue outlasts the ring --
// 14B: vectors staged at 6.5 us, the ring (110-125 KiB) full at 4.4-5 us, the loader idle for ~2 us of every k_att / k_ffn_rk launch
// (profiles/r04/timelines_14B_1B5.txt) -- the stream waits for the consumers.  The consumer waves that do NOT stage (4..6) therefore take
// the workgroup's FIRST groups (the loader's tail only moves over LEADING free groups: freeing groups 4..6 would buy nothing) out of
// the ring into registers as soon as they land, BEFORE the vectors are staged: three groups (45-75 KiB) more room for the loader.  So
// the wait for "staged" (ready()) sits inside the first iteration, between the take and the dot products, and the first group of
// wave w is (w + 3) mod 7.  The first group's epilogue inputs are still requested right in front of its dot products (an early
// scattered global load would sit in the CU's in-order return queue in front of the stream).
// Measured (profiles/r04/early_take_ab.txt, A/B on one box): 14B k_att 19.0 -> 18.0 us, token +1.5 %; at 3-4 KiB rows, where the ring
// covers the prologue anyway, it loses 0.5 % (7B) / 3 % (3B) -- so: on for rows >= 5 KiB.
#ifndef RWKV_EARLY_TAKE
#define RWKV_EARLY_TAKE 2       // 0 off, 1 on, 2 by row size
#endif
template <int S> __device__ __forceinline__ constexpr bool early_take() { return RWKV_EARLY_TAKE == 1 || (RWKV_EARLY_TAKE == 2 && S >= 5); }
// the consumer waves' streaming loop (wave < NC): same pre / epi contract as stream_groups; ready() = wait until the vectors are staged
// and fetch the scalars the epilogues need (called once, by every wave)
template <int R, int S, int PAT, class Pre, class Epi, class Ready>
__device__ __forceinline__ void ring_groups(int g0, int g1, int nu, const unsigned char *ring, GldsCtl *ctl, const unsigned *xq, int lane, int wave,
                                            int chunks, Pre pre, Epi epi, Ready ready, unsigned &fail, unsigned long long *g_tl_groups = nullptr)
{
#ifdef RWKV_TL_GROUPS
    // debug build (tools/timeline.py): where a consumer wave's time goes.  Stamps of the wave: 1 inputs of its first group requested,
    // 2 first group taken, 4 its dot products and reductions done, 3 its epilogue done (5 stays "staged", 6 / 7 the kernel's end)
    int rr = 0;
#endif
    constexpr int NWP = NT / 2 / 64;
    constexpr bool ET = early_take<S>();
    bool first = true;
    if (!ET) { ready(); first = false; }
    const bool late_pre = ET && wave >= NWP;       // the waves that take their first group before the vectors are staged
    for (int g = g0 + (ET ? (wave + NC - NWP) % NC : wave); g < g1; g += NC) {
        decltype(pre(g)) in;
        if (!(first && late_pre)) in = pre(g);
#ifdef RWKV_TL_GROUPS
        if (rr == 0) tl_stamp(g_tl_groups, 1);
#endif
        u32x4 w[R][S];
        glds_take<R, S>(w, g - g0, nu, ring, ctl, lane, fail);
        if (first) { ready(); if (late_pre) in = pre(g); first = false; }
#ifdef RWKV_TL_GROUPS
        asm volatile("" : "+v"(w[R - 1][S - 1]));
        if (rr < 1) tl_stamp(g_tl_groups, 2);
#endif
        unsigned long long T[R];
        group_dot<R, S, PAT, false>(w, xq, lane, T, nullptr, 0, chunks, false);
#ifdef RWKV_TL_GROUPS
        asm volatile("" : "+s"(T[R - 1]));
        if (rr < 1) tl_stamp(g_tl_groups, 4);
#endif
        epi(g, T, in);
#ifdef RWKV_TL_GROUPS
        if (rr < 1) tl_stamp(g_tl_groups, 3);
        rr++;
#endif
    }
    if (first) ready();      // a wave without a group still meets the others (time-outs are reported per wave)
}
// ring kernels: the control block is zeroed before the order barrier
__device__ __forceinline__ void ring_init(GldsCtl *gc)
{
    if (threadIdx.x < (unsigned)GLDS_CTL_ZERO) reinterpret_cast<unsigned *>(gc)[threadIdx.x] = 0u;
}
// LayerNorm-site prologue of a ring kernel, called by the consumer waves (wave < NC): waves 0..3 stage the NV vectors and
// publish the scalars, the others wait; contains the workgroup's order barrier, which the loader executes once as well
template <int NV, int S>
__device__ __forceinline__ void ring_site(const SiteStatic &st, const SiteDyn &dy, const double *x, int D, double *red, unsigned *xq,
                                          bool publish_stats, GldsCtl *gc, unsigned long long *tl, unsigned &fail)
{
    constexpr int NTP = NT / 2, NWP = NTP / 64, NQP = (S * 256 + NTP - 1) / NTP;
    const int nqd = D >> 2;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    if (wave_id() < NWP) {
        double tc[NV];
        float mc[NV];
#pragma unroll
        for (int m = 0; m < NV; m++) { tc[m] = st.TC[m]; mc[m] = st.maxC[m]; }
        SiteTuple tup;
        site_tuple_load(dy, tup);
        double xl[NQP][4];
        f32x4 Cq[NQP][NV], Bq[NQP][NV];
#pragma unroll
        for (int i = 0; i < NQP; i++) {
            const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
            load_quad_f64(x, qc, xl[i]);
#pragma unroll
            for (int m = 0; m < NV; m++) {
                Cq[i][m] = reinterpret_cast<const f32x4 *>(st.C + (size_t)m * D)[qc];
                Bq[i][m] = reinterpret_cast<const f32x4 *>(dy.B + (size_t)m * D)[qc];
            }
        }
        tl_stamp(tl, 1);
        if (threadIdx.x == 0) *spin = 0u;
        __syncthreads();   // order
        SiteRed<NV> r;
        site_reduce<NV, NWP>(st, dy, tup, D, red, r, tc, mc, tl, spin);
        if (publish_stats && blockIdx.x == 0 && threadIdx.x == 0) { dy.lnstat[0] = r.mean; dy.lnstat[1] = r.rstd; }
        tl_stamp(tl, 4);
        site_stage<NV, NQP, S, NTP>(xl, Cq, Bq, r, xq, nqd);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int m = 0; m < NV; m++) { bc[m] = (float)r.S[m]; bc[4 + m] = r.amax[m]; }
        }
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&gc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __syncthreads();   // order
    }
}
// ... and its second half, run by every consumer wave inside its first group (ring_groups' ready()): the vectors are staged; the
// scalars of the site come out of LDS
template <int NV>
__device__ __forceinline__ void ring_site_ready(double *red, GldsCtl *gc, SiteRed<NV> &sr, unsigned &fail, unsigned long long *tl)
{
    constexpr int NWP = NT / 2 / 64;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    wait_count(&gc->staged, NWP, fail);
#pragma unroll
    for (int m = 0; m < NV; m++) { sr.S[m] = (double)bc[m]; sr.amax[m] = bc[4 + m]; }
    sr.mean = sr.rstd = 0.0;
    tl_stamp(tl, 5);
}
// plain-vector prologue of a ring kernel (k_attout, k_ffnv), same roles
template <int NVEC, int S>
__device__ __forceinline__ void ring_vec(const float *vec, const double *partS, const float *partM, int n_part, int D, double *red, unsigned *xq,
                                         GldsCtl *gc, unsigned long long *tl, unsigned &fail)
{
    constexpr int XVD = xvd<S>();
    constexpr int NTP = NT / 2, NWP = NTP / 64, NQP = (S * 256 + NTP - 1) / NTP;
    const int lane = threadIdx.x & 63, wave = wave_id(), nqd = D >> 2;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    if (wave < NWP) {
        double ps = partS[(int)threadIdx.x < n_part ? threadIdx.x : 0];
        float pm = partM[(int)threadIdx.x < n_part ? threadIdx.x : 0];
        float vl[NVEC][NQP][4];
#pragma unroll
        for (int q = 0; q < NVEC; q++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
                const f32x4 t = reinterpret_cast<const f32x4 *>(vec + (size_t)q * D)[qc];
                vl[q][i][0] = t[0]; vl[q][i][1] = t[1]; vl[q][i][2] = t[2]; vl[q][i][3] = t[3];
            }
        tl_stamp(tl, 1);
        if (threadIdx.x == 0) *spin = 0u;
        __syncthreads();   // order
        if ((int)threadIdx.x >= n_part) { ps = 0.0; pm = 0.f; }
        float *redf = reinterpret_cast<float *>(red + RED_MAX);
        const double ws = wave_sum(ps);
        const float wm = wave_max(pm);
        if (lane == 0) { red[RED_OFFS + wave] = ws; redf[wave] = wm; }
        tl_stamp(tl, 3);
        if (lane == 0) __hip_atomic_fetch_add(spin, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        wait_count(spin, NWP, fail);
        double ts = 0.0; float tm = 0.f;
#pragma unroll
        for (int i = 0; i < NWP; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
        tl_stamp(tl, 4);
#pragma unroll
        for (int q = 0; q < NVEC; q++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP;
                if (qd < S * 256) stage_quad(xq + q * XVD, qd, vl[q][i], inv_scale(tm), qd < nqd);
            }
        if (threadIdx.x == 0) { bc[0] = (float)ts; bc[4] = tm; }
        if (lane == 0) __hip_atomic_fetch_add(&gc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __syncthreads();   // order
    }
}
__device__ __forceinline__ void ring_vec_ready(double *red, GldsCtl *gc, float &Sf, float &amax, unsigned &fail, unsigned long long *tl)
{
    constexpr int NWP = NT / 2 / 64;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    wait_count(&gc->staged, NWP, fail);
    Sf = bc[0]; amax = bc[4];
    tl_stamp(tl, 5);
}

// ------------------------------------------------------------------------------------------
struct FirstArgs {
    const float *embed;   // [V][D] f32 (device resident), first stage only
    const double *ln;     // layernorm table; rows 0,1 = ln0 weight, bias
    double *x;            // residual stream [D]: written here
    const double *x_in;   // later pipeline stage: the residual vector handed over by the previous stage (may be x itself)
    SiteStatic st;        // ln1 site of this stage's first layer
    SiteDyn dy;
    const double *sxy;    // state xy of that layer
    size_t slot_stride;
    const Ctl *ctl;
    int D;
    int from_token;       // 1: x = ln0(embed[token]) (rwkv.cu:513-524); 0: x was handed over by the previous stage
};
// Produce the residual vector (first stage) and open the first layer's ln1 site.  A few workgroups,
// each owning a slice of the channels (every one recomputes the ln0 statistics of the 16 KB row).
__global__ __launch_bounds__(NT) void k_first(FirstArgs a)
{
    __shared__ double red[RED_BYTES / 8];
    const int D = a.D;
    const int j0 = block_lo(D), j1 = block_hi(D);
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;
    double mean = 0.0, rstd = 1.0;
    const float *row = a.from_token ? a.embed + (size_t)a.ctl->token * D : nullptr;
    if (a.from_token) {
        double s[2] = {0.0, 0.0};
        for (int j = threadIdx.x; j < D; j += NT) { const double v = (double)row[j]; s[0] += v; s[1] += v * v; }
        block_sum<2>(s, red + RED_STATS);
        mean = s[0] / (double)D;
        rstd = 1.0 / sqrt((s[1] - s[0] * mean) / (double)(D - 1));
        __syncthreads();
    }
    SiteAcc<3> acc;
    acc.clear();
    for (int j = j0 + threadIdx.x; j < j1; j += NT) {
        double x;
        if (a.from_token) { x = a.ln[j] * (((double)row[j] - mean) * rstd) + a.ln[D + j]; a.x[j] = x; }
        else { x = a.x_in[j]; a.x[j] = x; }
        SitePre<3> pre;
        site_prefetch<3>(a.st, j, pre);
        site_emit<3>(pre, a.dy, D, j, x, a.sxy[so + j], acc);
    }
    site_publish<3, 64>(acc, a.dy, red);
}

// ------------------------------------------------------------------------------------------
struct AttArgs {
    const double *x;                      // residual stream [D]
    SiteStatic st;                        // ln1 site (3 vectors: k, v, r)
    SiteDyn dy;
    const uint8_t *w;                     // [D][3][D] u8: rows K_i, V_i, R_i of channel i
    const unsigned *rs;                   // [D][3] row sums of w (for the 2^22 limb offset)
    const double *uw, *ew;                // precomputed bonus+decay and exp(decay), [D]
    const float *r_att, *o_att;           // att_out scale / offset (to pre-scale the gated wkv)
    double *saa, *sbb;                    // state arrays [slots][L][D], already offset to this layer
    size_t slot_stride;                   // L*D
    float *ybuf;                          // [D] gated wkv * r_att (input vector of k_attout)
    double *partS;                        // [gridDim.x] partial sums of gated wkv * o_att
    float *partM;                         // [gridDim.x] partial max |ybuf| (k_attout's quantisation scale)
    const Ctl *ctl;
    int D;
    int ns;                               // ring kernels: LDS slots
    unsigned long long *tl;               // optional phase timeline (see tl_stamp)
    unsigned *herr;                       // ring kernels: the context's error word (raise_error)
};

// ln1 site -> K,V,R dequant-GEMV -> WKV (rwkv.cu:535-545; kernels :351-392, :58-100, :221-259)
struct AttIn { unsigned rs[3]; double aa, bb, uw, ew; float ra, oa; };
template <int S, int NB, int RING = 0>
__global__ __launch_bounds__(NT) void k_att(AttArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4;
    if constexpr (RING) RWKV_ARGS_NOW(a.x, a.st.C, a.st.TC, a.st.maxC, a.dy.B, a.dy.pd, a.dy.pf, a.dy.n_part);
    const int g0 = block_lo(D);
    const int g1 = block_hi(D);
    // (not on the loader wave: a compiler-visible load pending there makes hipcc guard the register it was given with s_waitcnt vmcnt(0)
    // wherever the loader's code happens to reuse it -- in round 5 that was inside the loader's loop, i.e. a drain of its DMA queue per group)
    size_t so = 0;
    if (!RING || wave != NC) so = (size_t)a.ctl->slot * a.slot_stride;

    tl_stamp(a.tl, 0);
    // every wave requests its first groups, even one without work (it re-reads a neighbour's rows): a
    // branch around the loads would make hipcc's waitcnt pass drain the weights early
    auto base = [&](int g) { return a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 3 * D; };
    double sc[3];
    float S0, S1, S2;
    auto scalars = [&](const SiteRed<3> &sr) {
        sc[0] = scale_of(sr.amax[0]); sc[1] = scale_of(sr.amax[1]); sc[2] = scale_of(sr.amax[2]);
        S0 = (float)sr.S[0]; S1 = (float)sr.S[1]; S2 = (float)sr.S[2];
    };
    // channel g = one group of three rows (K, V, R); the wave that finished them runs the WKV recurrence and the
    // receptance gate of that channel at once (rwkv.cu:242-255): no staging of k/v/r, no pass after the last row
    double part = 0.0;
    float pmax = 0.f;
    auto pre = [&](int g) {
        AttIn in;
#pragma unroll
        for (int m = 0; m < 3; m++) in.rs[m] = a.rs[g * 3 + m];
        in.aa = a.saa[so + g]; in.bb = a.sbb[so + g]; in.uw = a.uw[g]; in.ew = a.ew[g];
        in.ra = a.r_att[g]; in.oa = a.o_att[g];
        return in;
    };
    auto epi = [&](int g, const unsigned long long (&T)[3], const AttIn &in) {
        if (lane == 0) {
            const float k = row_value(T[0], in.rs[0], sc[0]) + S0, v = row_value(T[1], in.rs[1], sc[1]) + S1;
            const float r = row_value(T[2], in.rs[2], sc[2]) + S2;
            const double vv = (double)v;
            const double e1 = exp(in.uw + (double)k);
            double y = (in.aa + e1 * vv) / (in.bb + e1);
            y = (1.0 / (1.0 + (double)expf(-r))) * y;       // rwkv.cu:250: exp of a float argument
            const double ek = exp((double)k);
            a.saa[so + g] = (in.aa + ek * vv) * in.ew;
            a.sbb[so + g] = (in.bb + ek) * in.ew;
            const float yf = (float)y;                       // att_out GEMV casts its input to f32 (rwkv.cu:290)
            const float ys = yf * in.ra;
            a.ybuf[g] = ys;
            part += (double)(yf * in.oa);
            pmax = fmaxf(pmax, fabsf(ys));
        }
    };
    unsigned fail = 0u;        // a bounded wait of this wave gave up (wait_count)
    if constexpr (RING) {
        GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + RED_BYTES + 3 * S * 3072);
        unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
        if (wave == NC) {
            fail = glds_loader<3, S>(base, g0, g1, (size_t)D, chunks, a.ns, lds_addr(ring), gc, lane);
            tl_stamp(a.tl, 2);
        } else {
            ring_site<3, S>(a.st, a.dy, a.x, D, red, xq, true, gc, a.tl, fail);
            auto ready = [&]() { SiteRed<3> sr; ring_site_ready<3>(red, gc, sr, fail, a.tl); scalars(sr); };
            ring_groups<3, S, PAT_PER_ROW>(g0, g1, a.ns, ring, gc, xq, lane, wave, chunks, pre, epi, ready, fail, a.tl);
        }
    } else {
        u32x4 wA[3][S], wB[3][S];
        int gA, gB;
        unsigned *gctr = group_counter(red);
        first_groups<NB>(g0, wave, gctr, gA, gB);   // the counter is visible behind the prologue's barriers
        SiteRed<3> sr;
        site_open<3, 3, S, (RWKV_SPLIT & 1) != 0, NB>(a.st, a.dy, a.x, D, red, xq, wA, wB, base(gA), base(gB), (size_t)D, sr, true, a.tl);
        scalars(sr);
        stream_groups<3, S, PAT_PER_ROW, NB>(wA, wB, gA, gB, g0, g1, gctr, xq, lane, (size_t)D, chunks, base, pre, epi);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();   // every wave is past its last read of the reduction scratch
    block_sum_max(part, pmax, red + RED_PART);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part; a.partM[blockIdx.x] = pmax; }
    if constexpr (RING) ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ------------------------------------------------------------------------------------------
struct AttOutArgs {
    const uint8_t *w;      // [D][D] u8 rows = output channels
    const unsigned *rs;    // [D] row sums
    const float *ybuf;     // [D] pre-scaled input vector
    const double *partS;   // [n_part] partial offset sums (n_part <= NT)
    const float *partM;    // [n_part] partial max |ybuf|
    int n_part;
    double *x;             // residual stream, updated in place (row-owned)
    const double *lnw, *lnb;   // ln1 rows of this layer (f64): the state xy written here is ln1's output
    const double *lnstat;  // [2] mean, rstd of the ln1 site (published by k_att)
    double *sxy;           // state xy of this layer
    SiteStatic st;         // ln2 site (2 vectors) that this kernel opens for k_ffn_rk
    SiteDyn dy;
    const double *sdd;     // state dd of this layer (the ln2 site's prev)
    size_t slot_stride;
    const Ctl *ctl;
    int D;
    int ns;                // ring kernels: LDS slots
    unsigned long long *tl;
    unsigned *herr;
};

// att_out dequant-GEMV + residual through f32 (rwkv.cu:548-553), R rows per group; commits state xy;
// opens the ln2 site for the rows it owns
template <int R> struct AttOutIn { unsigned rsum; double xold, lw, lb, prev2; SitePre<2> pre; int mi, shift; };
template <int S, int R, int NB, int RING = 0>
__global__ __launch_bounds__(NT) void k_attout(AttOutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4;
    tl_stamp(a.tl, 0);
    if constexpr (RING) RWKV_ARGS_NOW(a.ybuf, a.partS, a.partM, a.n_part);
    const int G = (D + R - 1) / R;
    const int g0 = block_lo(G);
    const int g1 = block_hi(G);

    double mean1 = 0.0, rstd1 = 1.0;
    size_t so = 0;
    if (!RING || wave != NC) { mean1 = a.lnstat[0]; rstd1 = a.lnstat[1]; so = (size_t)a.ctl->slot * a.slot_stride; }      // (not on the loader wave: see k_att)
    auto base = [&](int gg) {
        int row = (gg < g1 ? gg : (g1 > g0 ? g1 - 1 : 0)) * R;
        if (row > D - R) row = D - R;          // the last group may overlap the previous one
        return a.w + (size_t)row * D;
    };
    float Sf, amax;
    double sc;
    SiteAcc<2> acc;
    acc.clear();
    auto pre = [&](int g) {
        AttOutIn<R> in;
        int row0 = g * R;
        in.shift = (row0 > D - R) ? row0 - (D - R) : 0;
        row0 -= in.shift;
        in.mi = row0 + (lane < R ? lane : 0);               // lane r owns row row0 + r
        in.rsum = a.rs[in.mi];
        in.xold = a.x[in.mi]; in.lw = a.lnw[in.mi]; in.lb = a.lnb[in.mi]; in.prev2 = a.sdd[so + in.mi];
        site_prefetch<2>(a.st, in.mi, in.pre);
        return in;
    };
    auto epi = [&](int, const unsigned long long (&T)[R], const AttOutIn<R> &in) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (lane == r && r >= in.shift) {
                const float accf = (float)in.xold + (row_value(T[r], in.rsum, sc) + Sf);   // f32 accumulator pre-loaded with x (:548)
                const double xnew = (double)accf;                                           // :553
                a.x[in.mi] = xnew;
                a.sxy[so + in.mi] = in.lw * ((in.xold - mean1) * rstd1) + in.lb;           // mixatt's state write (:385): ln1 output
                site_emit<2>(in.pre, a.dy, D, in.mi, xnew, in.prev2, acc);
            }
        }
    };
    unsigned fail = 0u;        // a bounded wait of this wave gave up (wait_count)
    if constexpr (RING) {
        GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + RED_BYTES + S * 3072);
        unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
        if (wave == NC) {
            fail = glds_loader<R, S>(base, g0, g1, (size_t)D, chunks, a.ns, lds_addr(ring), gc, lane);
            tl_stamp(a.tl, 2);
        } else {
            ring_vec<1, S>(a.ybuf, a.partS, a.partM, a.n_part, D, red, xq, gc, a.tl, fail);
            auto ready = [&]() { ring_vec_ready(red, gc, Sf, amax, fail, a.tl); sc = scale_of(amax); };
            ring_groups<R, S, PAT_SHARED>(g0, g1, a.ns, ring, gc, xq, lane, wave, chunks, pre, epi, ready, fail, nullptr);
        }
    } else {
        u32x4 wA[R][S], wB[R][S];
        int gA, gB;
        unsigned *gctr = group_counter(red);
        first_groups<NB>(g0, wave, gctr, gA, gB);
        vec_open<1, R, S, (RWKV_SPLIT & 2) != 0, NB>(a.ybuf, a.partS, a.partM, a.n_part, D, red, xq, wA, wB, base(gA), base(gB), (size_t)D, Sf, amax, a.tl);
        sc = scale_of(amax);
        stream_groups<R, S, PAT_SHARED, NB>(wA, wB, gA, gB, g0, g1, gctr, xq, lane, (size_t)D, chunks, base, pre, epi);
    }
    __syncthreads();   // every wave is past its last read of the reduction scratch (and of the staged vector / the ring)
    tl_stamp(a.tl, 6);
    site_publish<2, R>(acc, a.dy, xq);   // the staged vector is dead: its LDS is the scratch
    if constexpr (RING) ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ------------------------------------------------------------------------------------------
struct FfnRKArgs {
    const double *x;
    SiteStatic st;                    // ln2 site (2 vectors: ffn k, ffn r)
    SiteDyn dy;
    const uint8_t *w;                 // [D][5][D]: rows ffn_k out 4i..4i+3, then ffn_r out i
    const unsigned *rs;               // [D][5] row sums
    const float *r_fv, *o_fv;         // ffn_v scale / offset [4D]
    float *hbuf;                      // [4D] relu^2(k) * r_fv
    float *rgate;                     // [D] sigmoid(r)
    double *partS;                    // [gridDim.x]
    float *partM;                     // [gridDim.x] partial max |hbuf| (k_ffnv's quantisation scale)
    const Ctl *ctl;
    int D;
    int ns;                           // ring kernels: LDS slots
    unsigned long long *tl;           // optional phase timeline (see tl_stamp)
    unsigned *herr;
};

// ln2 site -> ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2 (rwkv.cu:557-573)
struct FfnRKIn { unsigned rsum; float rq, oq; };
template <int S, int NB, int RING = 0>
__global__ __launch_bounds__(NT) void k_ffn_rk(FfnRKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4;
    if constexpr (RING) RWKV_ARGS_NOW(a.x, a.st.C, a.st.TC, a.st.maxC, a.dy.B, a.dy.pd, a.dy.pf, a.dy.n_part);
    const int g0 = block_lo(D);
    const int g1 = block_hi(D);
    tl_stamp(a.tl, 0);

    auto base = [&](int g) { return a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 5 * D; };
    double sck, scr;
    float Sk, Sr;
    auto scalars = [&](const SiteRed<2> &sr) { sck = scale_of(sr.amax[0]); scr = scale_of(sr.amax[1]); Sk = (float)sr.S[0]; Sr = (float)sr.S[1]; };
    // channel g = the four ffn_k rows 4g..4g+3 and the ffn_r row g; lane q < 4 finishes hidden unit 4g + q (relu^2,
    // pre-scaled for ffn_v), lane 4 the receptance gate -- as soon as the group's sums exist
    double part = 0.0;
    float pmax = 0.f;
    auto pre = [&](int g) {
        FfnRKIn in;
        in.rsum = a.rs[g * 5 + (lane < 5 ? lane : 0)];
        const int kk = 4 * g + (lane < 4 ? lane : 0);
        in.rq = a.r_fv[kk]; in.oq = a.o_fv[kk];
        return in;
    };
    auto epi = [&](int g, const unsigned long long (&T)[5], const FfnRKIn &in) {
        float val = 0.f;     // lane r finishes row r (the sums are wave-uniform, the row sums per lane)
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const float vr = row_value(T[r], in.rsum, r < 4 ? sck : scr) + (r < 4 ? Sk : Sr);
            val = lane == r ? vr : val;
        }
        if (lane < 4) {
            float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
            h = h * h;
            const float hs = h * in.rq;
            a.hbuf[4 * g + lane] = hs;
            part += (double)(h * in.oq);
            pmax = fmaxf(pmax, fabsf(hs));
        } else if (lane == 4) {
            a.rgate[g] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
        }
    };
    unsigned fail = 0u;        // a bounded wait of this wave gave up (wait_count)
    if constexpr (RING) {
        GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + RED_BYTES + 2 * S * 3072);
        unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
        if (wave == NC) {
            fail = glds_loader<5, S>(base, g0, g1, (size_t)D, chunks, a.ns, lds_addr(ring), gc, lane);
            tl_stamp(a.tl, 2);
        } else {
            ring_site<2, S>(a.st, a.dy, a.x, D, red, xq, true, gc, a.tl, fail);
            auto ready = [&]() { SiteRed<2> sr; ring_site_ready<2>(red, gc, sr, fail, a.tl); scalars(sr); };
            ring_groups<5, S, PAT_FFN_RK>(g0, g1, a.ns, ring, gc, xq, lane, wave, chunks, pre, epi, ready, fail, a.tl);
        }
    } else {
        u32x4 wA[5][S], wB[5][S];
        int gA, gB;
        unsigned *gctr = group_counter(red);
        first_groups<NB>(g0, wave, gctr, gA, gB);
        SiteRed<2> sr;
        site_open<2, 5, S, (RWKV_SPLIT & 4) != 0, NB>(a.st, a.dy, a.x, D, red, xq, wA, wB, base(gA), base(gB), (size_t)D, sr, true, a.tl);
        scalars(sr);
        stream_groups<5, S, PAT_FFN_RK, NB>(wA, wB, gA, gB, g0, g1, gctr, xq, lane, (size_t)D, chunks, base, pre, epi);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();   // every wave is past its last read of the reduction scratch
    block_sum_max(part, pmax, red + RED_PART);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part; a.partM[blockIdx.x] = pmax; }
    if constexpr (RING) ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ------------------------------------------------------------------------------------------
struct FfnVArgs {
    const uint8_t *w;      // [D][4D] u8: row i = output channel i, as 4 quarter-rows of D bytes
    const unsigned *rs;    // [D] row sums (whole 4D row)
    const float *hbuf;     // [4D] pre-scaled hidden vector
    const double *partS;
    const float *partM;    // [n_part] partial max |hbuf|
    int n_part;
    const float *rgate;    // [D]
    double *x;             // residual stream (row-owned update)
    const double *lnw, *lnb;   // ln2 rows of this layer: the state dd written here is ln2's output
    const double *lnstat;  // [2] mean, rstd of the ln2 site (published by k_ffn_rk)
    double *sdd;
    SiteStatic st;         // the NEXT site: ln1 of layer l+1 (3 vectors) or ln_out -> head (1 vector)
    SiteDyn dy;
    const double *sprev;   // state xy of layer l+1 (prev of the next ln1 site); unused for the head site
    size_t slot_stride;
    const Ctl *ctl;
    int D;
    int ns;                // ring kernels: LDS slots
    unsigned long long *tl;
    unsigned *herr;
};

// ffn_v dequant-GEMV, x += v * sigmoid(r) (rwkv.cu:574-577); commits state dd; opens the next site
template <int NVN> struct FfnVIn { unsigned rsum; double xold, lw, lb, prevn; float rg; SitePre<NVN> pre; };
template <int S, int NVN, int NB, int RING = 0>
__global__ __launch_bounds__(NT) void k_ffnv(FfnVArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4;
    if constexpr (RING) RWKV_ARGS_NOW(a.hbuf, a.partS, a.partM, a.n_part);
    tl_stamp(a.tl, 0);
    const int g0 = block_lo(D);
    const int g1 = block_hi(D);

    double mean2 = 0.0, rstd2 = 1.0;
    size_t so = 0;
    if (!RING || wave != NC) { mean2 = a.lnstat[0]; rstd2 = a.lnstat[1]; so = (size_t)a.ctl->slot * a.slot_stride; }      // (not on the loader wave: see k_att)
    auto base = [&](int g) { return a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 4 * D; };
    float Sf, amax;   // one scale for the whole 4D hidden vector
    double sc;
    SiteAcc<NVN> acc;
    acc.clear();
    auto pre = [&](int g) {
        FfnVIn<NVN> in;
        in.rsum = a.rs[g];
        in.xold = a.x[g]; in.lw = a.lnw[g]; in.lb = a.lnb[g];
        in.prevn = NVN == 3 ? a.sprev[so + g] : 0.0;
        in.rg = a.rgate[g];
        site_prefetch<NVN>(a.st, g, in.pre);
        return in;
    };
    auto epi = [&](int g, const unsigned long long (&T)[4], const FfnVIn<NVN> &in) {
        if (lane == 0) {
            const float v = row_value((T[0] + T[1]) + (T[2] + T[3]), in.rsum, sc) + Sf;
            const double xnew = in.xold + (double)(v * in.rg);               // blockout, rwkv.cu:407 (f32 product)
            a.x[g] = xnew;
            a.sdd[so + g] = in.lw * ((in.xold - mean2) * rstd2) + in.lb;     // mixffn's state write (:344): ln2 output
            site_emit<NVN>(in.pre, a.dy, D, g, xnew, in.prevn, acc);
        }
    };
    unsigned fail = 0u;        // a bounded wait of this wave gave up (wait_count)
    if constexpr (RING) {
        GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + RED_BYTES + 4 * S * 3072);
        unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
        if (wave == NC) {
            fail = glds_loader<4, S>(base, g0, g1, (size_t)D, chunks, a.ns, lds_addr(ring), gc, lane);
            tl_stamp(a.tl, 2);
        } else {
            ring_vec<4, S>(a.hbuf, a.partS, a.partM, a.n_part, D, red, xq, gc, a.tl, fail);
            auto ready = [&]() { ring_vec_ready(red, gc, Sf, amax, fail, a.tl); sc = scale_of(amax); };
            ring_groups<4, S, PAT_PER_ROW>(g0, g1, a.ns, ring, gc, xq, lane, wave, chunks, pre, epi, ready, fail, a.tl);
        }
    } else {
        u32x4 wA[4][S], wB[4][S];
        int gA, gB;
        unsigned *gctr = group_counter(red);
        first_groups<NB>(g0, wave, gctr, gA, gB);
        vec_open<4, 4, S, (RWKV_SPLIT & 8) != 0, NB>(a.hbuf, a.partS, a.partM, a.n_part, D, red, xq, wA, wB, base(gA), base(gB), (size_t)D, Sf, amax, a.tl);
        sc = scale_of(amax);
        stream_groups<4, S, PAT_PER_ROW, NB>(wA, wB, gA, gB, g0, g1, gctr, xq, lane, (size_t)D, chunks, base, pre, epi);
    }
    __syncthreads();   // every wave is past its last read of the reduction scratch
    tl_stamp(a.tl, 6);
    site_publish<NVN, 1>(acc, a.dy, xq);   // the staged vector is dead: its LDS is the scratch
    if constexpr (RING) ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ------------------------------------------------------------------------------------------
struct HeadArgs {
    const double *x;
    SiteStatic st;             // ln_out site (1 vector, no token shift)
    SiteDyn dy;
    const uint8_t *w;          // [V][D] u8
    const unsigned *rs;        // [V] row sums
    float *logits;             // [max_ctx][V]
    float *blk_val;            // [gridDim.x] per-workgroup max logit (index 0 banned)
    unsigned *blk_idx;         // [gridDim.x]
    const Ctl *ctl;
    int D;
    int ns;                    // ring kernels: LDS slots
    unsigned *herr;
};

// ln_out site -> head dequant-GEMV -> logits (rwkv.cu:585-589); also per-workgroup argmax partials
template <int R> struct HeadIn { unsigned rsr[R]; int row0, shift; };
template <int S, int NB, int RING = 0, int R = RING ? RWKV_HEAD_RR : RWKV_HEAD_R>
__global__ __launch_bounds__(NT) void k_head(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    float *bval = reinterpret_cast<float *>(xq + XVD);
    unsigned *bidx = reinterpret_cast<unsigned *>(bval + NW);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int chunks = D >> 4;
    const int V = (int)VOCAB;
    const int G = (V + R - 1) / R;
    const int g0 = block_lo(G);
    const int g1 = block_hi(G);
    float *lg = a.logits;                                  // read at entry: behind the prologue's barriers it is a cold load (not on the loader wave: see k_att)
    if (!RING || wave != NC) lg = a.logits + (size_t)a.ctl->out_row * V;

    auto base = [&](int gg) {
        int row = (gg < g1 ? gg : (g1 > g0 ? g1 - 1 : 0)) * R;
        if (row > V - R) row = V - R;
        return a.w + (size_t)row * D;
    };
    float Sf;
    double sc;
    float best = -INFINITY;
    unsigned besti = 0xffffffffu;
    auto pre = [&](int g) {
        HeadIn<R> in;
        in.row0 = g * R;
        in.shift = (in.row0 > V - R) ? in.row0 - (V - R) : 0;
        in.row0 -= in.shift;
#pragma unroll
        for (int r = 0; r < R; r++) in.rsr[r] = a.rs[in.row0 + r];
        return in;
    };
    auto epi = [&](int, const unsigned long long (&T)[R], const HeadIn<R> &in) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = in.row0 + r;
            const float val = row_value(T[r], in.rsr[r], sc) + Sf;
            if (lane == r && r >= in.shift) lg[i] = val;
            if (i != 0 && (val > best || (val == best && (unsigned)i < besti))) { best = val; besti = (unsigned)i; }
        }
    };
    unsigned fail = 0u;        // a bounded wait of this wave gave up (wait_count)
    if constexpr (RING) {
        GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + RED_BYTES + S * 3072 + 64);   // behind bval / bidx
        unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
        if (wave == NC) {
            fail = glds_loader<R, S>(base, g0, g1, (size_t)D, chunks, a.ns, lds_addr(ring), gc, lane);
        } else {
            ring_site<1, S>(a.st, a.dy, a.x, D, red, xq, false, gc, nullptr, fail);
            auto ready = [&]() { SiteRed<1> sr; ring_site_ready<1>(red, gc, sr, fail, nullptr); Sf = (float)sr.S[0]; sc = scale_of(sr.amax[0]); };
            ring_groups<R, S, PAT_SHARED>(g0, g1, a.ns, ring, gc, xq, lane, wave, chunks, pre, epi, ready, fail);
        }
    } else {
        u32x4 wA[R][S], wB[R][S];
        int gA, gB;
        unsigned *gctr = group_counter(red);
        first_groups<NB>(g0, wave, gctr, gA, gB);
        SiteRed<1> sr;
        site_open<1, R, S, (RWKV_SPLIT & 16) != 0, NB>(a.st, a.dy, a.x, D, red, xq, wA, wB, base(gA), base(gB), (size_t)D, sr, false, nullptr);
        Sf = (float)sr.S[0]; sc = scale_of(sr.amax[0]);
        stream_groups<R, S, PAT_SHARED, NB>(wA, wB, gA, gB, g0, g1, gctr, xq, lane, (size_t)D, chunks, base, pre, epi);
    }
    if (lane == 0) { bval[wave] = best; bidx[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NW; i++)
            if (bval[i] > best || (bval[i] == best && bidx[i] < besti)) { best = bval[i]; besti = bidx[i]; }
        a.blk_val[blockIdx.x] = best;
        a.blk_idx[blockIdx.x] = besti;
    }
    if constexpr (RING) ring_report(fail, a.herr);
}

// finish the greedy pick: argmax over workgroup partials (ties -> lowest id), feed it back as the
// next token and append it to the generated-id list.  One 64-lane wave.
__global__ void k_argmax_finish(const float *blk_val, const unsigned *blk_idx, int n, Ctl *ctl,
                                unsigned long long *gen, unsigned gen_cap)
{
    float best = -INFINITY;
    unsigned besti = 0xffffffffu;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float v = blk_val[i];
        const unsigned id = blk_idx[i];
        if (v > best || (v == best && id < besti)) { best = v; besti = id; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const unsigned oi = __shfl_xor(besti, m, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (threadIdx.x == 0) {
        // all-NaN / all -inf logits leave no winner (the reference's un-stabilised WKV can overflow with unusual
        // weights): feed token 0 like the sampler's fallback instead of indexing the embedding out of bounds
        if (besti >= VOCAB) besti = 0u;
        const unsigned st = ctl->step;
        if (st < gen_cap) gen[st] = besti;
        ctl->token = besti;
        ctl->step = st + 1;
    }
}

// ------------------------------------------------------------------------------------------
// Load-time re-tile: file layout src[N][M] (output index k contiguous, rwkv.cu:290 indexing
// w[j*M + k]) -> row-per-output dst[row(k)][N] with row(k) = (k/G)*RS + k%G + off.
__global__ void k_retile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int N, int M,
                         int G, int RS, int off)
{
    __shared__ uint8_t tile[64][65];
    const int k0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 64 x 4
    for (int jj = ty; jj < 64; jj += 4) {
        const int j = j0 + jj, k = k0 + tx;
        tile[jj][tx] = (j < N && k < M) ? src[(size_t)j * M + k] : 0;
    }
    __syncthreads();
    for (int kk = ty; kk < 64; kk += 4) {
        const int k = k0 + kk, j = j0 + tx;
        if (k < M && j < N) {
            const size_t row = (size_t)(k / G) * RS + (k % G) + off;
            dst[row * N + j] = tile[tx][kk];
        }
    }
}

// row sums of a re-tiled matrix: rs[row] = sum_j w_t[row][j]; one wave per row (load time)
__global__ void k_rowsum(const uint8_t *__restrict__ w_t, unsigned *__restrict__ rs, size_t rows, int N)
{
    const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(w_t + row * (size_t)N);
    unsigned acc = 0;
    for (int c = lane; c < (N >> 4); c += 64) {
        const u32x4 v = p[c];
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_udot4(v[q], 0x01010101u, acc, false);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) rs[row] = acc;
}

// uw = bonus + decay, ew = exp(decay)   (constants of rwkv.cu:247-252, hoisted out of the token loop)
__global__ void k_prep_wkv(const double *decay, const double *bonus, double *uw, double *ew, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uw[i] = bonus[i] + decay[i]; ew[i] = exp(decay[i]); }
}

// Load-time tables of vector m of one LayerNorm site (see "LayerNorm sites" above): consumer side
// C[m][j] = r mix lnw; producer side P[j][PW] holds BL = r mix lnb, BP = r (1 - mix) and the same
// with the offset o in place of the scale r (Co, BoL, BoP) at [m*5 + 0..4].  mix == nullptr means
// no token shift (ln_out -> head).  Products are formed in f64 and stored as f32.  TC[m] = sum of the
// stored Co, maxC[m] = max |C|.  One workgroup per call.
__global__ __launch_bounds__(NT) void k_site_static(const double *lnw, const double *lnb, const double *mix, const float *r, const float *o,
                                                    float *C, float *P, double *TC, float *maxC, int m, int PW, int D)
{
    __shared__ double red[RED_BYTES / 8];
    double s[1] = {0.0};
    float mx[1] = {0.f};
    for (int j = threadIdx.x; j < D; j += NT) {
        const double mk = mix ? mix[j] : 1.0, rr = (double)r[j], oo = (double)o[j];
        const float c = (float)(rr * mk * lnw[j]), co = (float)(oo * mk * lnw[j]);
        C[(size_t)m * D + j] = c;
        float *pp = P + (size_t)j * PW + m * 5;
        pp[0] = (float)(rr * mk * lnb[j]);
        pp[1] = (float)(rr * (1.0 - mk));
        pp[2] = co;
        pp[3] = (float)(oo * mk * lnb[j]);
        pp[4] = (float)(oo * (1.0 - mk));
        s[0] += (double)co;
        mx[0] = fmaxf(mx[0], fabsf(c));
    }
    block_sum<1>(s, red + RED_STATS);
    block_max<1>(mx, red + RED_MAX);
    if (threadIdx.x == 0) { TC[m] = s[0]; maxC[m] = mx[0]; }
}

} // namespace rwkvk
