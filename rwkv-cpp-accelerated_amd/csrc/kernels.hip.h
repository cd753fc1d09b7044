// kernels.hip.h -- gfx950 (MI355X / CDNA4) device code of the RWKV-v4 uint8 engine.
//
// One token = 4 launches per layer + embed + head (+ argmax), replacing the ~28 stream
// operations per layer of the reference driver (rwkv.cu:528-581):
//
//   k_embed_ln0   rwkv.cu:513-524   embedding row gather (device-resident table) + ln0
//   k_att         rwkv.cu:535-545   ln1 + token-shift mix + K/V/R dequant-GEMV + WKV recurrence
//   k_attout      rwkv.cu:548-553   att_out dequant-GEMV + residual (+ commits state xy)
//   k_ffn_rk      rwkv.cu:557-573   ln2 + mix + ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2
//   k_ffnv        rwkv.cu:574-577   ffn_v dequant-GEMV, x += v * sigmoid(r) (+ commits state dd)
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
//    24-bit fixed point (3 unsigned byte limbs, scale = max|.|/8388000) and staged in LDS, and
//    the u8 x u8 products run on v_dot4_u32_u8 (4 MACs per lane-instruction, exact u32
//    accumulation, no byte->float converts).  Measured on MI355X the f32 formulation
//    (v_cvt_f32_ubyteN + v_fma_f32 per weight byte) cost as much VALU time as the HBM stream
//    itself; the limb form needs 12 instructions per 16 weight bytes instead of 32 and its
//    rounding error (<= 1.2e-7 max|x|) is below that of an f32 FMA chain.  The offset term is
//    one scalar per vector.
//  * Every GEMV needs its complete input vector, so LayerNorm / token-shift prologues are
//    recomputed per workgroup from the D-vector (L2 resident).  Prologue inputs are requested
//    first, then the first steps of the weight stream, so the prologue runs under the stream.
//  * Grid = one 512-thread workgroup per CU (8 waves); each wave keeps R*S x 16 B loads in flight
//    and refills a step's registers with the next row group as soon as the step is consumed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkvk {

constexpr int NT = 512;          // threads per workgroup
constexpr int NW = NT / 64;      // wavefronts per workgroup
constexpr int RED_BYTES = 1024;  // LDS scratch for workgroup reductions
constexpr unsigned VOCAB = 50277u;
constexpr float QLIM = 8388000.0f;       // |quantised activation| <= QLIM < 2^23
constexpr double QOFF = 8388608.0;       // 2^23: limbs hold q + 2^23 as an unsigned 24-bit number

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
// outside the row sums used for the 2^23 offset correction).
__device__ __forceinline__ void stage_quad(unsigned *xq, int qd, const float (&xr)[4], float inv_s, bool real)
{
    unsigned d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const unsigned qi = (unsigned)(__float2int_rn(xr[e] * inv_s) + 8388608);
        d0 |= (qi & 255u) << (8 * e);
        d1 |= ((qi >> 8) & 255u) << (8 * e);
        d2 |= ((qi >> 16) & 255u) << (8 * e);
    }
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
// (q' = 24-bit unsigned limb value), every lane gets all R sums.
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

// ------------------------------------------------------------------------------------------
// Prologue discipline.  vmcnt retires loads IN ORDER, so a prologue load issued after the weight
// loads would not return before every weight byte of the wave has landed.  Each kernel therefore
// issues, in program order: (1) ALL of its prologue inputs, straight-line and branch-free
// (out-of-range elements clamp their index), (2) the first steps of the first row group, and only
// then (3) consumes the prologue inputs -- hipcc's counted s_waitcnt vmcnt(N) lets the LayerNorm /
// mix / quantise / LDS staging run while the weight loads are in flight.  The static per-channel
// parameters are packed at load time into float4 tables (k_pack_*).  A thread owns QUADS of 4
// consecutive elements (quad qd = tid + i*NT), which is what one limb dword holds.

// LayerNorm statistics of a D-vector held as NQ quads per thread (reference semantics: mean =
// sum/D, variance over D-1, no epsilon -- rwkv.cu:40-57,412-450), in f64: one workgroup reduction
// of (sum, sum of squares); var = (sumsq - sum*mean)/(D-1).
template <int NQ>
__device__ __forceinline__ void ln_stats(const double (&xl)[NQ][4], int D, double &mean, double &rstd, double *red)
{
    double s[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const bool real = (int)(threadIdx.x + i * NT) * 4 < D;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double v = real ? xl[i][e] : 0.0;
            s[0] += v;
            s[1] += v * v;
        }
    }
    block_sum<2>(s, red + RED_STATS);
    mean = s[0] / (double)D;
    const double var = (s[1] - s[0] * mean) / (double)(D - 1);
    rstd = 1.0 / sqrt(var);
}
__device__ __forceinline__ void load_quad_f64(const double *p, int qd, double (&out)[4])
{
    const f64x2 a = reinterpret_cast<const f64x2 *>(p)[qd * 2], b = reinterpret_cast<const f64x2 *>(p)[qd * 2 + 1];
    out[0] = a[0]; out[1] = a[1]; out[2] = b[0]; out[3] = b[1];
}

// ------------------------------------------------------------------------------------------
struct EmbedArgs {
    const float *embed;   // [V][D] f32 (device resident)
    const double *ln;     // layernorm table; rows 0,1 = ln0 weight, bias
    double *x;            // residual stream [D]
    const Ctl *ctl;
    int D;
};
// rwkv.cu:513-524.  One workgroup.
__global__ __launch_bounds__(NT) void k_embed_ln0(EmbedArgs a)
{
    __shared__ double red[NW * 2];
    const int D = a.D;
    const float *row = a.embed + (size_t)a.ctl->token * D;
    double s[2] = {0.0, 0.0};
    for (int j = threadIdx.x; j < D; j += NT) { const double v = (double)row[j]; s[0] += v; s[1] += v * v; }
    block_sum<2>(s, red);
    const double mean = s[0] / (double)D;
    const double rstd = 1.0 / sqrt((s[1] - s[0] * mean) / (double)(D - 1));
    for (int j = threadIdx.x; j < D; j += NT)
        a.x[j] = a.ln[j] * (((double)row[j] - mean) * rstd) + a.ln[D + j];
}

// ------------------------------------------------------------------------------------------
struct AttArgs {
    const double *x;                      // residual stream [D]
    const f32x4 *pk;                      // [3][4][D/4] packed {lnw,lnb,mixk,mixv | mixr,rk,rv,rr | ok,ov,or,0}, see k_pack_att
    const uint8_t *w;                     // [D][3][D] u8: rows K_i, V_i, R_i of channel i
    const unsigned *rs;                   // [D][3] row sums of w (for the 2^23 limb offset)
    const double *uw, *ew;                // precomputed bonus+decay and exp(decay), [D]
    const float *r_att, *o_att;           // att_out scale / offset (to pre-scale the gated wkv)
    double *sxy, *saa, *sbb;              // state arrays [slots][L][D], already offset to this layer
    size_t slot_stride;                   // L*D
    double *xx_buf;                       // [D] ln1 output, committed to sxy by k_attout
    float *ybuf;                          // [D] gated wkv * r_att (input vector of k_attout)
    double *partS;                        // [gridDim.x] partial sums of gated wkv * o_att
    float *partM;                         // [gridDim.x] partial max |ybuf| (k_attout's quantisation scale)
    const Ctl *ctl;
    int D;
};

// ln1 -> mix -> K,V,R dequant-GEMV -> WKV (rwkv.cu:535-545; kernels :351-392, :58-100, :221-259)
template <int S>
__global__ __launch_bounds__(NT) void k_att(AttArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = nquads<S>();
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    float *stash = reinterpret_cast<float *>(xq + 3 * XVD);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;

    // (1) prologue inputs
    double xl[NQ][4], pv[NQ][4];
    f32x4 P[NQ][4][3];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
        load_quad_f64(a.x, qc, xl[i]);
        load_quad_f64(a.sxy + so, qc, pv[i]);
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int q = 0; q < 3; q++) P[i][e][q] = a.pk[(q * 4 + e) * nqd + qc];   // [piece][elem-in-quad][quad]: lane-contiguous
    }
    // (2) first steps of the first group's weights.  Unconditional (a wave without a group re-reads
    // a neighbour's rows): a branch around the loads would make hipcc's waitcnt pass assume the
    // no-load path and drain the weights early.
    u32x4 w[3][S];
    int g = g0 + wave;
    const uint8_t *wb = a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 3 * D;
    group_load<3, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);

    // (3) LayerNorm, token-shift mix, pre-scale by the per-row quantisation scale
    double mean, rstd;
    ln_stats<NQ>(xl, D, mean, rstd, red);
    double Ssum[3] = {0.0, 0.0, 0.0};
    float amax[3] = {0.f, 0.f, 0.f};
    float xr[NQ][3][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        const bool real = qd < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double xx = (double)P[i][e][0][0] * ((xl[i][e] - mean) * rstd) + (double)P[i][e][0][1];
            const double prev = pv[i][e];
            const double mk = (double)P[i][e][0][2], mv = (double)P[i][e][0][3], mr = (double)P[i][e][1][0];
            const float fk = (float)(mk * xx + (1.0 - mk) * prev);   // rwkv.cu:382-384: rounded to f32
            const float fv = (float)(mv * xx + (1.0 - mv) * prev);
            const float fr = (float)(mr * xx + (1.0 - mr) * prev);
            xr[i][0][e] = fk * P[i][e][1][1]; xr[i][1][e] = fv * P[i][e][1][2]; xr[i][2][e] = fr * P[i][e][1][3];
            if (real) {
                Ssum[0] += (double)(fk * P[i][e][2][0]); Ssum[1] += (double)(fv * P[i][e][2][1]); Ssum[2] += (double)(fr * P[i][e][2][2]);
#pragma unroll
                for (int m = 0; m < 3; m++) amax[m] = fmaxf(amax[m], fabsf(xr[i][m][e]));
                if (qd * 4 + e >= g0 && qd * 4 + e < g1) a.xx_buf[qd * 4 + e] = xx;   // every workgroup publishes the channels it owns (not one straggler all of them)
            }
        }
    }
    block_max<3>(amax, red + RED_MAX);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        if (qd < S * 256) {
#pragma unroll
            for (int m = 0; m < 3; m++) stage_quad(xq + m * XVD, qd, xr[i][m], inv_scale(amax[m]), qd < nqd);
        }
    }
    __syncthreads();                                                       // staged vectors visible
    // the bulk of the first group is requested only now: its issue stalls on the full memory pipe,
    // and a stall in front of a workgroup barrier would make every wave wait for the slowest one
    group_load<3, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);
    const double sc[3] = {scale_of(amax[0]), scale_of(amax[1]), scale_of(amax[2])};

    for (; g < g1; g += NW) {
        unsigned long long T[3];
        const bool nv = g + NW < g1;
        // per-group epilogue inputs are requested BEFORE the dot issues the refill loads: a load
        // placed after them would, by in-order vmcnt, wait for the whole next group to land
        const unsigned rsum = a.rs[g * 3 + (lane < 3 ? lane : 0)];
        group_dot<3, S, PAT_PER_ROW>(w, xq, lane, T, a.w + (size_t)(nv ? g + NW : 0) * 3 * D, (size_t)D, chunks, nv);
#pragma unroll
        for (int m = 0; m < 3; m++)
            if (lane == m) stash[(g - g0) * 3 + m] = row_value(T[m], rsum, sc[m]);
    }
    block_sum<3>(Ssum, red + RED_OFFS);   // offset terms: only needed by the epilogue (also the barrier before it)

    // WKV recurrence + receptance gate, one lane per channel (rwkv.cu:242-255)
    double part[1] = {0.0};
    float pmax[1] = {0.f};
    if ((int)threadIdx.x < g1 - g0) {
        const int i = g0 + threadIdx.x;
        const float k = stash[threadIdx.x * 3 + 0] + (float)Ssum[0], v = stash[threadIdx.x * 3 + 1] + (float)Ssum[1],
                    r = stash[threadIdx.x * 3 + 2] + (float)Ssum[2];
        const double aa = a.saa[so + i], bb = a.sbb[so + i];
        const double vv = (double)v;
        const double e1 = exp(a.uw[i] + (double)k);
        double y = (aa + e1 * vv) / (bb + e1);
        y = (1.0 / (1.0 + (double)expf(-r))) * y;       // rwkv.cu:250: exp of a float argument
        const double ek = exp((double)k), ew = a.ew[i];
        a.saa[so + i] = (aa + ek * vv) * ew;
        a.sbb[so + i] = (bb + ek) * ew;
        const float yf = (float)y;                       // att_out GEMV casts its input to f32 (rwkv.cu:290)
        const float ys = yf * a.r_att[i];
        a.ybuf[i] = ys;
        part[0] = (double)(yf * a.o_att[i]);
        pmax[0] = fabsf(ys);
    }
    block_sum<1>(part, red + RED_PART);
    block_max<1>(pmax, red + RED_AUX);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part[0]; a.partM[blockIdx.x] = pmax[0]; }
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
    const double *xx_buf;  // ln1 output of this token -> new state xy
    double *sxy;           // state xy of this layer
    size_t slot_stride;
    const Ctl *ctl;
    int D;
};

// att_out dequant-GEMV + residual through f32 (rwkv.cu:548-553), R rows per group
template <int S, int R>
__global__ __launch_bounds__(NT) void k_attout(AttOutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NQ = nquads<S>();
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int G = (D + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    float yl[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
        const f32x4 t = reinterpret_cast<const f32x4 *>(a.ybuf)[qc];
        yl[i][0] = t[0]; yl[i][1] = t[1]; yl[i][2] = t[2]; yl[i][3] = t[3];
    }
    double Ssum[1] = {a.partS[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};
    float amax[1] = {a.partM[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;

    u32x4 w[R][S];
    int g = g0 + wave;
    auto rowbase = [&](int gg) {
        int row = gg * R;
        if (row > D - R) row = D - R;
        return a.w + (size_t)row * D;
    };
    const uint8_t *wb = rowbase(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0));
    group_load<R, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);

    // scale and offset come from the producer's per-workgroup partials: one reduction round
    if ((int)threadIdx.x >= a.n_part) { Ssum[0] = 0.0; amax[0] = 0.f; }
    {
        float *redf = reinterpret_cast<float *>(red + RED_MAX);
        const double ws = wave_sum(Ssum[0]);
        const float wm = wave_max(amax[0]);
        if (lane == 0) { red[RED_OFFS + wave] = ws; redf[wave] = wm; }
        __syncthreads();
        double ts = 0.0; float tm = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
        Ssum[0] = ts; amax[0] = tm;
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        if (qd < S * 256) stage_quad(xq, qd, yl[i], inv_scale(amax[0]), qd < nqd);
    }
    __syncthreads();   // staged vector visible
    group_load<R, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);   // bulk issue after the last barrier (see k_att)
    const float Sf = (float)Ssum[0];
    const double sc = scale_of(amax[0]);

    for (; g < g1; g += NW) {
        unsigned long long T[R];
        const bool nv = g + NW < g1;
        int row0 = g * R;
        const int shift = (row0 > D - R) ? row0 - (D - R) : 0;   // last group may overlap the previous one
        row0 -= shift;
        // epilogue inputs first (see k_att): lane r owns row row0 + r
        const int mi = row0 + (lane < R ? lane : 0);
        const unsigned rsum = a.rs[mi];
        const double xold = a.x[mi], xxn = a.xx_buf[mi];
        group_dot<R, S, PAT_SHARED>(w, xq, lane, T, rowbase(nv ? g + NW : 0), (size_t)D, chunks, nv);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (lane == r && r >= shift) {
                const float acc = (float)xold + (row_value(T[r], rsum, sc) + Sf);   // f32 accumulator pre-loaded with x (:548)
                a.x[mi] = (double)acc;                                              // :553
                a.sxy[so + mi] = xxn;                                               // mixatt's state write (:385), deferred
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
struct FfnRKArgs {
    const double *x;
    const f32x4 *pk;                  // [2][4][D/4] packed {lnw,lnb,mixk,mixr | rk,ok,rr,or}, see k_pack_ffn
    const uint8_t *w;                 // [D][5][D]: rows ffn_k out 4i..4i+3, then ffn_r out i
    const unsigned *rs;               // [D][5] row sums
    const float *r_fv, *o_fv;         // ffn_v scale / offset [4D]
    const double *sdd;                // state dd of this layer (read only here)
    size_t slot_stride;
    double *xx_buf;                   // [D] ln2 output -> committed to sdd by k_ffnv
    float *hbuf;                      // [4D] relu^2(k) * r_fv
    float *rgate;                     // [D] sigmoid(r)
    double *partS;                    // [gridDim.x]
    float *partM;                     // [gridDim.x] partial max |hbuf| (k_ffnv's quantisation scale)
    const Ctl *ctl;
    int D;
    unsigned long long *tl;           // optional phase timeline (see tl_stamp)
};

// ln2 -> mix -> ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2 (rwkv.cu:557-573)
template <int S>
__global__ __launch_bounds__(NT) void k_ffn_rk(FfnRKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = nquads<S>();
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    float *stash = reinterpret_cast<float *>(xq + 2 * XVD);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;
    tl_stamp(a.tl, 0);

    double xl[NQ][4], pv[NQ][4];
    f32x4 P[NQ][4][2];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
        load_quad_f64(a.x, qc, xl[i]);
        load_quad_f64(a.sdd + so, qc, pv[i]);
#pragma unroll
        for (int e = 0; e < 4; e++) { P[i][e][0] = a.pk[e * nqd + qc]; P[i][e][1] = a.pk[(4 + e) * nqd + qc]; }   // lane-contiguous
    }
    u32x4 w[5][S];
    int g = g0 + wave;
#ifdef RWKV_EXP_SAMEROWS   // timing experiment only: waves stream 64 cache-resident groups -> compute-only time
#define RWKV_EXP_G(gg) ((gg) % 64)
#else
#define RWKV_EXP_G(gg) (gg)
#endif
    const uint8_t *wb = a.w + (size_t)RWKV_EXP_G(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 5 * D;
    group_load<5, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);
    tl_stamp(a.tl, 1);

    double mean, rstd;
    ln_stats<NQ>(xl, D, mean, rstd, red);
    tl_stamp(a.tl, 2);
    double Ssum[2] = {0.0, 0.0};
    float amax[2] = {0.f, 0.f};
    float xr[NQ][2][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        const bool real = qd < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double xx = (double)P[i][e][0][0] * ((xl[i][e] - mean) * rstd) + (double)P[i][e][0][1];
            const double prev = pv[i][e];
            const double mk = (double)P[i][e][0][2], mr = (double)P[i][e][0][3];
            const float fk = (float)(mk * xx + (1.0 - mk) * prev);   // f64 mix (:341-342), f32 cast in the GEMV (:290)
            const float fr = (float)(mr * xx + (1.0 - mr) * prev);
            xr[i][0][e] = fk * P[i][e][1][0]; xr[i][1][e] = fr * P[i][e][1][2];
            if (real) {
                Ssum[0] += (double)(fk * P[i][e][1][1]); Ssum[1] += (double)(fr * P[i][e][1][3]);
                amax[0] = fmaxf(amax[0], fabsf(xr[i][0][e])); amax[1] = fmaxf(amax[1], fabsf(xr[i][1][e]));
                if (qd * 4 + e >= g0 && qd * 4 + e < g1) a.xx_buf[qd * 4 + e] = xx;   // every workgroup publishes the channels it owns (not one straggler all of them)
            }
        }
    }
    block_max<2>(amax, red + RED_MAX);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        if (qd < S * 256) {
            stage_quad(xq, qd, xr[i][0], inv_scale(amax[0]), qd < nqd);
            stage_quad(xq + XVD, qd, xr[i][1], inv_scale(amax[1]), qd < nqd);
        }
    }
    __syncthreads();
    tl_stamp(a.tl, 3);
    group_load<5, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);   // bulk issue after the last barrier (see k_att)
    const double sc[2] = {scale_of(amax[0]), scale_of(amax[1])};
    bool first_ = true;

    for (; g < g1; g += NW) {
        unsigned long long T[5];
        const bool nv = g + NW < g1;
        const unsigned rsum = a.rs[g * 5 + (lane < 5 ? lane : 0)];   // before the refills (see k_att)
        group_dot<5, S, PAT_FFN_RK>(w, xq, lane, T, a.w + (size_t)RWKV_EXP_G(nv ? g + NW : 0) * 5 * D, (size_t)D, chunks, nv);
#pragma unroll
        for (int r = 0; r < 5; r++)
            if (lane == r) stash[(g - g0) * 5 + r] = row_value(T[r], rsum, sc[r < 4 ? 0 : 1]);
        if (first_) { tl_stamp(a.tl, 4); first_ = false; }
    }
    block_sum<2>(Ssum, red + RED_OFFS);   // offset terms: only needed by the epilogue (also the barrier before it)
    tl_stamp(a.tl, 5);

    double part[1] = {0.0};
    float pmax[1] = {0.f};
    for (int t = threadIdx.x; t < 5 * (g1 - g0); t += NT) {
        const int q = t % 5, i = g0 + t / 5;
        const float val = stash[t] + (float)Ssum[q < 4 ? 0 : 1];
        if (q < 4) {
            float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
            h = h * h;
            const int kk = 4 * i + q;
            const float hs = h * a.r_fv[kk];
            a.hbuf[kk] = hs;
            part[0] += (double)(h * a.o_fv[kk]);
            pmax[0] = fmaxf(pmax[0], fabsf(hs));
        } else {
            a.rgate[i] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
        }
    }
    block_sum<1>(part, red + RED_PART);
    block_max<1>(pmax, red + RED_AUX);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part[0]; a.partM[blockIdx.x] = pmax[0]; }
    tl_stamp(a.tl, 6);
}

// ------------------------------------------------------------------------------------------
// Wave-specialised variant: loader waves + LDS ring.
//
// Measured on MI355X (tools/timeline.py): in the all-waves-do-everything kernels above the weight
// stream and the prologue fight for the same in-order waves -- a wave that has asked for more loads
// than the memory pipe accepts stalls AT ISSUE and cannot run its share of the prologue, and a
// wave that asks for little leaves HBM idle for the ~6 us the prologue takes.  Here the roles are
// split inside the workgroup: NL loader waves do nothing but copy the workgroup's (contiguous)
// weight region HBM -> registers -> LDS ring, one row group per ring slot, two groups in flight per
// loader; the other NC waves run the prologue and then drain the ring with the same dot4 code.
// The ring fills while the prologue runs, so the stream never waits for it.
// No s_barrier after the role split (it would need the loaders): the consumers synchronise among
// themselves through LDS counters; every spin is bounded.
constexpr int NL = 2;           // loader waves
constexpr int NC = NW - NL;     // consumer waves
constexpr int NCT = NC * 64;    // consumer threads
struct RingCtl {
    unsigned ready[8];          // ready[slot] = generation whose data is complete in the slot
    unsigned done[8];           // done[slot]  = generation the consumers have finished reading
    unsigned bar;               // consumer barrier arrivals
    unsigned err;               // a bounded spin gave up
    unsigned go;                // consumer waves whose prologue loads have landed (loaders start at NC)
    unsigned pad[13];
};
__device__ __forceinline__ bool spin_ge(unsigned *p, unsigned want, unsigned *err)
{
    for (int i = 0; i < (1 << 22); i++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return false;
}
// barrier among the NC consumer waves only (phase counts the barriers executed so far)
__device__ __forceinline__ void cbar(RingCtl *rc, unsigned &phase)
{
    phase++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&rc->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    spin_ge(&rc->bar, phase * NC, &rc->err);
}
// reductions among the consumer waves; `red` must be a scratch region that is not reused by a
// later reduction before every wave has read it (callers pass distinct regions), so one barrier each
template <int K>
__device__ __forceinline__ void csum(double (&v)[K], double *red, RingCtl *rc, unsigned &phase, int cw)
{
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[cw * K + k] = v[k];
    }
    cbar(rc, phase);
#pragma unroll
    for (int k = 0; k < K; k++) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NC; i++) t += red[i * K + k];
        v[k] = t;
    }
}
template <int K>
__device__ __forceinline__ void cmax(float (&v)[K], double *redd, RingCtl *rc, unsigned &phase, int cw)
{
    float *red = reinterpret_cast<float *>(redd);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_max(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[cw * K + k] = v[k];
    }
    cbar(rc, phase);
#pragma unroll
    for (int k = 0; k < K; k++) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NC; i++) t = fmaxf(t, red[i * K + k]);
        v[k] = t;
    }
}

// loader wave `lw`: copies groups lw, lw+NL, ... (gbytes each, contiguous from `base`) into the ring.
// NLD = 1 KiB wave-loads per group.  Two register sets: the next group's loads are in flight while
// the current one is written to LDS.  The issue is branch-free (see group_dot's refill).
template <int NLD, int NSLOT>
__device__ __forceinline__ void ring_loader(const uint8_t *__restrict__ base, unsigned gbytes, int ng, unsigned char *ring,
                                            RingCtl *rc, int lw, int lane)
{
    u32x4 A[NLD], B[NLD];
    auto issue = [&](u32x4 (&buf)[NLD], int gi) {
        const bool valid = gi < ng;
        const uint8_t *gb = base + (size_t)(valid ? gi : 0) * gbytes;
        const unsigned mask = valid ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            unsigned off = (unsigned)(i * 1024 + lane * 16);
            off = off < gbytes ? off : gbytes - 16;
            buf[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(gb + (off & mask)));
        }
    };
    auto drain = [&](u32x4 (&buf)[NLD], int gi) {
        const int slot = gi % NSLOT;
        const unsigned gen = (unsigned)(gi / NSLOT) + 1u;
        if (gen > 1u) spin_ge(&rc->done[slot], gen - 1u, &rc->err);   // previous tenant fully consumed
        unsigned char *dst = ring + (size_t)slot * gbytes;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const unsigned off = (unsigned)(i * 1024 + lane * 16);
            if (off < gbytes) *reinterpret_cast<u32x4 *>(dst + off) = buf[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&rc->ready[slot], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int gi = lw;
    issue(A, gi);
    while (gi < ng) {
        issue(B, gi + NL);
        drain(A, gi);
        gi += NL;
        if (gi >= ng) break;
        issue(A, gi + NL);
        drain(B, gi);
        gi += NL;
    }
}

// consumer: fetch the R*S pieces of the group in ring slot `slot` into registers
template <int R, int S>
__device__ __forceinline__ void ring_fetch(u32x4 (&w)[R][S], const unsigned char *slot, int D, int chunks, int lane)
{
#pragma unroll
    for (int s = 0; s < S; s++) {
        int c = lane + 64 * s;
        c = c < chunks ? c : chunks - 1;
#pragma unroll
        for (int r = 0; r < R; r++) w[r][s] = *reinterpret_cast<const u32x4 *>(slot + (size_t)r * D + ((unsigned)c << 4));
    }
}

// ln2 -> mix -> ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2 (rwkv.cu:557-573): loader/consumer variant
template <int S, int NSLOT>
__global__ __launch_bounds__(NT) void k_ffn_rk_ring(FfnRKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = (S * 256 + NCT - 1) / NCT;      // quads per consumer thread
    RingCtl *rc = reinterpret_cast<RingCtl *>(smem);
    double *red = reinterpret_cast<double *>(smem + 128);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + 128 + RED_BYTES);
    float *stash = reinterpret_cast<float *>(xq + 2 * XVD);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    const int ng = g1 - g0;
    const int gpb = (D + gridDim.x - 1) / gridDim.x + 1;
    unsigned char *ring = reinterpret_cast<unsigned char *>(stash + ((gpb * 5 + 3) & ~3));
    const unsigned gbytes = 5u * (unsigned)D;

    if (threadIdx.x < 32) reinterpret_cast<unsigned *>(rc)[threadIdx.x] = 0u;
    __syncthreads();   // the only full-workgroup barrier: ring control words are zero
    tl_stamp(a.tl, 0);

    if (wave < NL) {   // ---------------- loader waves ----------------
        // The CU's vector-memory pipe is shared: prologue loads (L2 hits) issued behind a saturating
        // HBM stream see HBM-like latency (measured: staging 6 -> 11 us).  So the stream starts
        // only when every consumer wave has RECEIVED its prologue inputs (~2 us), and then runs
        // under the prologue's arithmetic.
        spin_ge(&rc->go, NC, &rc->err);
        ring_loader<5 * S, NSLOT>(a.w + (size_t)g0 * gbytes, gbytes, ng, ring, rc, wave, lane);
        return;
    }
    // ---------------- consumer waves ----------------
    const int cw = wave - NL, ct = threadIdx.x - NL * 64;
    unsigned phase = 0;
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;
    double xl[NQ][4], pv[NQ][4];
    f32x4 P[NQ][4][2];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = ct + i * NCT, qc = qd < nqd ? qd : nqd - 1;
        load_quad_f64(a.x, qc, xl[i]);
        load_quad_f64(a.sdd + so, qc, pv[i]);
#pragma unroll
        for (int e = 0; e < 4; e++) { P[i][e][0] = a.pk[e * nqd + qc]; P[i][e][1] = a.pk[(4 + e) * nqd + qc]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prologue inputs are in registers: release the loaders
    if (lane == 0) __hip_atomic_fetch_add(&rc->go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    tl_stamp(a.tl, 1);
    double st[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const bool real = (ct + i * NCT) < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) { const double v = real ? xl[i][e] : 0.0; st[0] += v; st[1] += v * v; }
    }
    csum<2>(st, red, rc, phase, cw);
    const double mean = st[0] / (double)D;
    const double rstd = 1.0 / sqrt((st[1] - st[0] * mean) / (double)(D - 1));
    tl_stamp(a.tl, 2);
    double Ssum[2] = {0.0, 0.0};
    float amax[2] = {0.f, 0.f};
    float xr[NQ][2][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = ct + i * NCT;
        const bool real = qd < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double xx = (double)P[i][e][0][0] * ((xl[i][e] - mean) * rstd) + (double)P[i][e][0][1];
            const double prev = pv[i][e];
            const double mk = (double)P[i][e][0][2], mr = (double)P[i][e][0][3];
            const float fk = (float)(mk * xx + (1.0 - mk) * prev);
            const float fr = (float)(mr * xx + (1.0 - mr) * prev);
            xr[i][0][e] = fk * P[i][e][1][0]; xr[i][1][e] = fr * P[i][e][1][2];
            if (real) {
                Ssum[0] += (double)(fk * P[i][e][1][1]); Ssum[1] += (double)(fr * P[i][e][1][3]);
                amax[0] = fmaxf(amax[0], fabsf(xr[i][0][e])); amax[1] = fmaxf(amax[1], fabsf(xr[i][1][e]));
                if (qd * 4 + e >= g0 && qd * 4 + e < g1) a.xx_buf[qd * 4 + e] = xx;   // every workgroup publishes the channels it owns (not one straggler all of them)
            }
        }
    }
    cmax<2>(amax, red + 16, rc, phase, cw);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = ct + i * NCT;
        if (qd < S * 256) {
            stage_quad(xq, qd, xr[i][0], inv_scale(amax[0]), qd < nqd);
            stage_quad(xq + XVD, qd, xr[i][1], inv_scale(amax[1]), qd < nqd);
        }
    }
    cbar(rc, phase);   // staged vectors visible to all consumers
    tl_stamp(a.tl, 3);
    const double sc[2] = {scale_of(amax[0]), scale_of(amax[1])};
    bool first_ = true;

    for (int gi = cw; gi < ng; gi += NC) {
        const int slot = gi % NSLOT;
        const unsigned gen = (unsigned)(gi / NSLOT) + 1u;
        const unsigned rsum = a.rs[(g0 + gi) * 5 + (lane < 5 ? lane : 0)];
        spin_ge(&rc->ready[slot], gen, &rc->err);
        u32x4 w[5][S];
        ring_fetch<5, S>(w, ring + (size_t)slot * gbytes, D, chunks, lane);
        unsigned long long T[5];
        group_dot<5, S, PAT_FFN_RK, false>(w, xq, lane, T, nullptr, 0, chunks, false);
        // every lane's ring reads were consumed by the dots above: the slot may be refilled
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&rc->done[slot], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 5; r++)
            if (lane == r) stash[gi * 5 + r] = row_value(T[r], rsum, sc[r < 4 ? 0 : 1]);
        if (first_) { tl_stamp(a.tl, 4); first_ = false; }
    }
    csum<2>(Ssum, red + 32, rc, phase, cw);   // offset terms + barrier: all stashes written
    tl_stamp(a.tl, 5);

    double part[1] = {0.0};
    float pmax[1] = {0.f};
    for (int t = ct; t < 5 * ng; t += NCT) {
        const int q = t % 5, i = g0 + t / 5;
        const float val = stash[t] + (float)Ssum[q < 4 ? 0 : 1];
        if (q < 4) {
            float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
            h = h * h;
            const int kk = 4 * i + q;
            const float hs = h * a.r_fv[kk];
            a.hbuf[kk] = hs;
            part[0] += (double)(h * a.o_fv[kk]);
            pmax[0] = fmaxf(pmax[0], fabsf(hs));
        } else {
            a.rgate[i] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
        }
    }
    csum<1>(part, red + 48, rc, phase, cw);
    cmax<1>(pmax, red + 56, rc, phase, cw);
    if (ct == 0) { a.partS[blockIdx.x] = part[0]; a.partM[blockIdx.x] = pmax[0]; }
    tl_stamp(a.tl, 6);
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
    const double *xx_buf;  // ln2 output -> new state dd
    double *sdd;
    size_t slot_stride;
    const Ctl *ctl;
    int D;
};

// ffn_v dequant-GEMV, x += v * sigmoid(r) (rwkv.cu:574-577)
template <int S>
__global__ __launch_bounds__(NT) void k_ffnv(FfnVArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = nquads<S>();
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);

    float hl[4][NQ][4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
            const f32x4 t = reinterpret_cast<const f32x4 *>(a.hbuf + (size_t)q * D)[qc];
            hl[q][i][0] = t[0]; hl[q][i][1] = t[1]; hl[q][i][2] = t[2]; hl[q][i][3] = t[3];
        }
    double Ssum[1] = {a.partS[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};
    float amax[1] = {a.partM[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};   // one scale for the whole 4D vector
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;

    u32x4 w[4][S];
    int g = g0 + wave;
    const uint8_t *wb = a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 4 * D;
    group_load<4, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);

    if ((int)threadIdx.x >= a.n_part) { Ssum[0] = 0.0; amax[0] = 0.f; }
    {   // scale and offset from the producer's per-workgroup partials: one reduction round
        float *redf = reinterpret_cast<float *>(red + RED_MAX);
        const double ws = wave_sum(Ssum[0]);
        const float wm = wave_max(amax[0]);
        if (lane == 0) { red[RED_OFFS + wave] = ws; redf[wave] = wm; }
        __syncthreads();
        double ts = 0.0; float tm = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
        Ssum[0] = ts; amax[0] = tm;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int qd = threadIdx.x + i * NT;
            if (qd < S * 256) stage_quad(xq + q * XVD, qd, hl[q][i], inv_scale(amax[0]), qd < nqd);
        }
    __syncthreads();   // staged vector visible
    group_load<4, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);   // bulk issue after the last barrier (see k_att)
    const float Sf = (float)Ssum[0];
    const double sc = scale_of(amax[0]);

    for (; g < g1; g += NW) {
        unsigned long long T[4];
        const bool nv = g + NW < g1;
        const unsigned rsum = a.rs[g];   // epilogue inputs before the refills (see k_att)
        const double xold = a.x[g], xxn = a.xx_buf[g];
        const float rg = a.rgate[g];
        group_dot<4, S, PAT_PER_ROW>(w, xq, lane, T, a.w + (size_t)(nv ? g + NW : 0) * 4 * D, (size_t)D, chunks, nv);
        if (lane == 0) {
            const float v = row_value((T[0] + T[1]) + (T[2] + T[3]), rsum, sc) + Sf;
            a.x[g] = xold + (double)(v * rg);   // blockout, rwkv.cu:407 (f32 product)
            a.sdd[so + g] = xxn;                 // mixffn's state write (:344), deferred
        }
    }
}

// ------------------------------------------------------------------------------------------
struct HeadArgs {
    const double *x;
    const f32x4 *pk;           // [4][D/4] packed {lnw, lnb, r, o} of ln_out / head, see k_pack_head
    const uint8_t *w;          // [V][D] u8
    const unsigned *rs;        // [V] row sums
    float *logits;             // [max_ctx][V]
    float *blk_val;            // [gridDim.x] per-workgroup max logit (index 0 banned)
    unsigned *blk_idx;         // [gridDim.x]
    const Ctl *ctl;
    int D;
};

// ln_out -> head dequant-GEMV -> logits (rwkv.cu:585-589); also per-workgroup argmax partials
template <int S>
__global__ __launch_bounds__(NT) void k_head(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = nquads<S>();
    constexpr int R = 4;
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    float *bval = reinterpret_cast<float *>(xq + XVD);
    unsigned *bidx = reinterpret_cast<unsigned *>(bval + NW);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4, nqd = D >> 2;
    const int V = (int)VOCAB;
    const int G = (V + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    double xl[NQ][4];
    f32x4 P[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
        load_quad_f64(a.x, qc, xl[i]);
#pragma unroll
        for (int e = 0; e < 4; e++) P[i][e] = a.pk[e * nqd + qc];   // lane-contiguous
    }
    u32x4 w[R][S];
    int g = g0 + wave;
    auto rowbase = [&](int gg) {
        int row = gg * R;
        if (row > V - R) row = V - R;
        return a.w + (size_t)row * D;
    };
    const uint8_t *wb = rowbase(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0));
    group_load<R, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);
    double mean, rstd;
    ln_stats<NQ>(xl, D, mean, rstd, red);
    double Ssum[1] = {0.0};
    float amax[1] = {0.f};
    float xr[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const bool real = (int)(threadIdx.x + i * NT) < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float f = (float)((double)P[i][e][0] * ((xl[i][e] - mean) * rstd) + (double)P[i][e][1]);
            xr[i][e] = f * P[i][e][2];
            if (real) { Ssum[0] += (double)(f * P[i][e][3]); amax[0] = fmaxf(amax[0], fabsf(xr[i][e])); }
        }
    }
    block_max<1>(amax, red + RED_MAX);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NT;
        if (qd < S * 256) stage_quad(xq, qd, xr[i], inv_scale(amax[0]), qd < nqd);
    }
    block_sum<1>(Ssum, red + RED_OFFS);
    group_load<R, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);   // bulk issue after the last barrier (see k_att)
    const float Sf = (float)Ssum[0];
    const double sc = scale_of(amax[0]);
    float *lg = a.logits + (size_t)a.ctl->out_row * V;

    float best = -INFINITY;
    unsigned besti = 0xffffffffu;
    for (; g < g1; g += NW) {
        unsigned long long T[R];
        const bool nv = g + NW < g1;
        int row0 = g * R;
        const int shift = (row0 > V - R) ? row0 - (V - R) : 0;
        row0 -= shift;
        const u32x4 rs4 = {a.rs[row0], a.rs[row0 + 1], a.rs[row0 + 2], a.rs[row0 + 3]};   // before the refills (see k_att)
        group_dot<R, S, PAT_SHARED>(w, xq, lane, T, rowbase(nv ? g + NW : 0), (size_t)D, chunks, nv);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = row0 + r;
            const float val = row_value(T[r], rs4[r], sc) + Sf;
            if (lane == r && r >= shift) lg[i] = val;
            if (i != 0 && (val > best || (val == best && (unsigned)i < besti))) { best = val; besti = (unsigned)i; }
        }
    }
    if (lane == 0) { bval[wave] = best; bidx[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NW; i++)
            if (bval[i] > best || (bval[i] == best && bidx[i] < besti)) { best = bval[i]; besti = bidx[i]; }
        a.blk_val[blockIdx.x] = best;
        a.blk_idx[blockIdx.x] = besti;
    }
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
        const unsigned st = ctl->step;
        if (st < gen_cap) gen[st] = besti;
        ctl->token = besti;
        ctl->step = st + 1;
    }
}

// ------------------------------------------------------------------------------------------
// Stand-alone dequant-GEMV on the same row engine (unit tests; the kernel behind
// cudac_mm8_one, rwkv.cu:267-311).  w_t is the re-tiled [M][N] matrix, rs its row sums.
// N <= 5120: rows whole; otherwise N = 4*Dq and rows are processed as 4 quarter-rows (ffn_v shape).
struct Mm8Args {
    const uint8_t *w_t;
    const unsigned *rs;
    const float *x, *r, *o;
    float *y;
    int N, M;
};
template <int S, bool QUARTERS>
__global__ __launch_bounds__(NT) void k_mm8(Mm8Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XVD = xvd<S>();
    constexpr int NQ = nquads<S>();
    constexpr int R = 4;
    constexpr int NVQ = QUARTERS ? 4 : 1;
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Dq = QUARTERS ? a.N / 4 : a.N;
    const int chunks = Dq >> 4, nqd = Dq >> 2;
    const int M = a.M;
    const int G = QUARTERS ? M : (M + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    double Ssum[1] = {0.0};
    float amax[1] = {0.f};
    float xr[NVQ][NQ][4];
#pragma unroll
    for (int q = 0; q < NVQ; q++)
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int qd = threadIdx.x + i * NT;
            const bool real = qd < nqd;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = q * Dq + (real ? qd : 0) * 4 + e;
                const float f = a.x[j];
                xr[q][i][e] = f * a.r[j];
                if (real) { Ssum[0] += (double)(f * a.o[j]); amax[0] = fmaxf(amax[0], fabsf(xr[q][i][e])); }
            }
        }
    block_max<1>(amax, red + RED_MAX);
#pragma unroll
    for (int q = 0; q < NVQ; q++)
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int qd = threadIdx.x + i * NT;
            if (qd < S * 256) stage_quad(xq + q * XVD, qd, xr[q][i], inv_scale(amax[0]), qd < nqd);
        }
    block_sum<1>(Ssum, red + RED_OFFS);
    const float Sf = (float)Ssum[0];
    const double sc = scale_of(amax[0]);

    for (int g = g0 + wave; g < g1; g += NW) {
        u32x4 w[R][S];
        unsigned long long T[R];
        if (QUARTERS) {
            group_load<R, S, 0, S>(w, a.w_t + (size_t)g * a.N, (size_t)Dq, chunks, lane);
            group_dot<R, S, PAT_PER_ROW>(w, xq, lane, T, a.w_t, 0, chunks, false);
            if (lane == 0) a.y[g] = row_value((T[0] + T[1]) + (T[2] + T[3]), a.rs[g], sc) + Sf;
        } else {
            int row0 = g * R;
            const int shift = (row0 > M - R) ? row0 - (M - R) : 0;
            row0 -= shift;
            group_load<R, S, 0, S>(w, a.w_t + (size_t)row0 * a.N, (size_t)a.N, chunks, lane);
            group_dot<R, S, PAT_SHARED>(w, xq, lane, T, a.w_t, 0, chunks, false);
#pragma unroll
            for (int r = 0; r < R; r++)
                if (lane == r && r >= shift) a.y[row0 + r] = row_value(T[r], a.rs[row0 + r], sc) + Sf;
        }
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

// Load-time packing of the static per-channel parameters into float4 tables.  A prologue thread
// owns quads of 4 consecutive channels (quad qd = j/4, e = j%4), so the tables are laid out
// [piece q][e][qd]: for a given (q, e) the 64 lanes of a wave read 64 consecutive float4 = one
// fully coalesced 1 KiB load.  The f64 tensors involved (layernorm rows, time-mix vectors) hold
// f32-representable values in converted checkpoints (converter: .double() of f32 tensors,
// convert_model.py:44-56), so the narrowing is exact there.
__global__ void k_pack_att(f32x4 *pk, const double *lnw, const double *lnb, const double *mk, const double *mv,
                           const double *mr, const float *rk, const float *rv, const float *rr,
                           const float *ok, const float *ov, const float *orr, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const int nqd = D >> 2, qd = j >> 2, e = j & 3;
    pk[(0 * 4 + e) * nqd + qd] = f32x4{(float)lnw[j], (float)lnb[j], (float)mk[j], (float)mv[j]};
    pk[(1 * 4 + e) * nqd + qd] = f32x4{(float)mr[j], rk[j], rv[j], rr[j]};
    pk[(2 * 4 + e) * nqd + qd] = f32x4{ok[j], ov[j], orr[j], 0.f};
}
__global__ void k_pack_ffn(f32x4 *pk, const double *lnw, const double *lnb, const double *mk, const double *mr,
                           const float *rk, const float *ok, const float *rr, const float *orr, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const int nqd = D >> 2, qd = j >> 2, e = j & 3;
    pk[(0 * 4 + e) * nqd + qd] = f32x4{(float)lnw[j], (float)lnb[j], (float)mk[j], (float)mr[j]};
    pk[(1 * 4 + e) * nqd + qd] = f32x4{rk[j], ok[j], rr[j], orr[j]};
}
__global__ void k_pack_head(f32x4 *pk, const double *lnw, const double *lnb, const float *r, const float *o, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const int nqd = D >> 2, qd = j >> 2, e = j & 3;
    pk[e * nqd + qd] = f32x4{(float)lnw[j], (float)lnb[j], r[j], o[j]};
}

} // namespace rwkvk
