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
// Design (DESIGN.md has the long form):
//  * The uint8 matrices are re-tiled at load into ROW-PER-OUTPUT order (w_t[k][j], j contiguous),
//    so one wavefront owns whole output rows: each lane streams 16-byte pieces of the row with
//    non-temporal global_load_dwordx4 (1 KiB per wave-instruction, every 128-B line used once),
//    accumulates 4 independent f32 FMA chains and the row is finished by a 64-lane shuffle
//    reduction.  No float atomics, no pre-zeroed accumulators, no cross-workgroup reduction:
//    results are deterministic and every epilogue (WKV, sigmoid, relu^2, residual) fuses into
//    the kernel that finished the row.
//  * sum_j x_j (u_jk r_j + o_j) = sum_j (x_j r_j) u_jk + sum_j x_j o_j : the activation vector is
//    pre-multiplied by the per-row scale once per workgroup and staged in LDS (chunk-interleaved
//    so every ds_read_b128 is conflict-free); the offset term is one scalar per vector.
//  * Every GEMV needs its complete input vector, so LayerNorm / token-shift prologues are
//    recomputed per workgroup from the D-vector (L2 resident) while the first weight loads --
//    issued before the prologue, they do not depend on it -- are already in flight.
//  * Grid = one 512-thread workgroup per CU (8 waves, up to 20 x 16 B loads in flight per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkvk {

constexpr int NT = 512;          // threads per workgroup
constexpr int NW = NT / 64;      // wavefronts per workgroup
constexpr int RED_BYTES = 1024;  // LDS scratch for workgroup reductions
constexpr unsigned VOCAB = 50277u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per-token control block, lives in device memory so a captured hipGraph can be replayed for
// any token / state slot / logits row (and so the device-side argmax can feed the next step)
struct Ctl {
    unsigned long long token;
    unsigned int slot;     // state slot (PARRALEL mode: sequence index; GPT mode: 0)
    unsigned int out_row;  // logits row
    unsigned int step;     // greedy-decode step counter
    unsigned int pad;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// workgroup-wide sum of K doubles; every thread gets the totals.  `red` = NW*K doubles of LDS.
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
    __syncthreads();
}

// LDS position of element j of an activation vector: 16-element chunk c = j/16 belongs to lane
// c%64 at step c/64; the four float4 pieces of a chunk are spread so that piece q of all 64
// lanes is contiguous (ds_read_b128 with lane-consecutive 16-B addresses: conflict-free).
__device__ __forceinline__ int xperm(int j)
{
    const int c = j >> 4, q = (j >> 2) & 3, e = j & 3;
    return ((((c >> 6) * 4 + q) * 64 + (c & 63)) << 2) + e;
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

// Row-group streaming.  A group = R rows `stride` bytes apart, each `chunks` 16-byte pieces long;
// piece c belongs to lane c%64 at step c/64, so one step of one row is a 1 KiB coalesced
// non-temporal wave load.  Lanes past the end of a row (rows that are not a multiple of 1 KiB)
// re-read the row's last piece instead of branching: the staged vector is zero there, and a
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

__device__ __forceinline__ void dot16(const u32x4 w, const f32x4 (&x)[4], float (&a)[4])
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const unsigned u = w[q];
        a[0] = fmaf((float)(u & 0xffu), x[q][0], a[0]);
        a[1] = fmaf((float)((u >> 8) & 0xffu), x[q][1], a[1]);
        a[2] = fmaf((float)((u >> 16) & 0xffu), x[q][2], a[2]);
        a[3] = fmaf((float)(u >> 24), x[q][3], a[3]);
    }
}

// Dot products of the loaded group with the staged vector(s); every lane gets all R sums.
//  * The LDS reads of the activation pieces are software-pipelined one (step, vector) item ahead
//    and pinned with sched barriers, so at most two 16-float pieces are live.
//  * Refill: as soon as step s of this group has been consumed its registers are re-loaded with
//    step s of the NEXT group (`next`), so the wave keeps R*S loads in flight across groups
//    instead of draining and restarting the memory pipe at every group boundary.  The refill is
//    unconditional and branch-free (a branch would make hipcc's waitcnt pass merge the two paths
//    and wait for the refill itself; two template copies in sibling branches get their common
//    byte->float converts hoisted and spilled): after a wave's LAST group, `next_valid` = false
//    degrades the refill to R*S loads of one and the same 16-byte piece (one L1-resident line),
//    which nobody waits for.
template <int R, int S, int PAT>
__device__ __forceinline__ void group_dot(u32x4 (&w)[R][S], const float *xv, int xvlen, int lane, float (&out)[R],
                                          const uint8_t *__restrict__ next, size_t stride, int chunks, bool next_valid)
{
    constexpr int NV = nvec<PAT, R>();
    constexpr int NI = S * NV;
    const unsigned mask = next_valid ? 0xffffffffu : 0u;
    const size_t nstride = next_valid ? stride : 0;
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
    f32x4 x[2][4];
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(xv) + lane;
#pragma unroll
    for (int q = 0; q < 4; q++) x[0][q] = xb[q * 64];
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int s = it / NV, v = it % NV;
        if (it + 1 < NI) {
            const int s1 = (it + 1) / NV, v1 = (it + 1) % NV;
            const f32x4 *p = xb + (v1 * (xvlen >> 2)) + (s1 * 4) * 64;
#pragma unroll
            for (int q = 0; q < 4; q++) x[(it + 1) & 1][q] = p[q * 64];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (xsel<PAT>(r) == v) {
                dot16(w[r][s], x[it & 1], acc[r]);
                // pin the partial sums here: without it LLVM sinks all FMAs below all LDS reads and
                // the whole staged vector becomes live at once (spills); the sched barrier keeps the
                // next row's 16 byte->float converts from being hoisted above this row's FMAs
                asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]));
                __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (v == NV - 1) {
            step_load<R, S>(w, s, next, nstride, chunks, lane, mask);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) out[r] = wave_sum((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]));
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
// (out-of-range elements clamp their index), (2) the first row group's weight loads, and only
// then (3) consumes the prologue inputs -- hipcc's counted s_waitcnt vmcnt(N) lets the LayerNorm /
// mix / LDS staging run while the weight loads are still in flight.  The static per-channel
// parameters are packed at load time into float4 tables (k_pack_*), so a prologue element costs
// 1-3 16-byte loads instead of up to 13 scalar ones.

// LayerNorm statistics of a D-vector held E elements per thread (reference semantics: mean =
// sum/D, variance over D-1, no epsilon -- rwkv.cu:40-57,412-450), in f64.
template <int E>
__device__ __forceinline__ void ln_stats(const double (&xl)[E], int D, double &mean, double &rstd, double *red)
{
    // one workgroup reduction of (sum, sum of squares) in f64; var = (sumsq - sum^2/D)/(D-1).
    // (f64 leaves ~1e-13 relative after the cancellation for |mean| up to 1e2 sigma.)
    double s[2] = {0.0, 0.0};
#pragma unroll
    for (int e = 0; e < E; e++) {
        const double v = (threadIdx.x + e * NT < D) ? xl[e] : 0.0;
        s[0] += v;
        s[1] += v * v;
    }
    block_sum<2>(s, red);
    mean = s[0] / (double)D;
    const double var = (s[1] - s[0] * mean) / (double)(D - 1);
    rstd = 1.0 / sqrt(var);
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
    double s[1] = {0.0};
    for (int j = threadIdx.x; j < D; j += NT) s[0] += (double)row[j];
    block_sum<1>(s, red);
    const double mean = s[0] / (double)D;
    double q[1] = {0.0};
    for (int j = threadIdx.x; j < D; j += NT) { const double d = (double)row[j] - mean; q[0] += d * d; }
    block_sum<1>(q, red);
    const double stdv = sqrt(q[0] / (double)(D - 1));
    for (int j = threadIdx.x; j < D; j += NT)
        a.x[j] = a.ln[j] * (((double)row[j] - mean) / stdv) + a.ln[D + j];
}

// ------------------------------------------------------------------------------------------
struct AttArgs {
    const double *x;                      // residual stream [D]
    const f32x4 *pk;                      // [D][3] packed {lnw,lnb,mixk,mixv | mixr,rk,rv,rr | ok,ov,or,0}
    const uint8_t *w;                     // [D][3][D] u8: rows K_i, V_i, R_i of channel i
    const double *uw, *ew;                // precomputed bonus+decay and exp(decay), [D]
    const float *r_att, *o_att;           // att_out scale / offset (to pre-scale the gated wkv)
    double *sxy, *saa, *sbb;              // state arrays [slots][L][D], already offset to this layer
    size_t slot_stride;                   // L*D
    double *xx_buf;                       // [D] ln1 output, committed to sxy by k_attout
    float *ybuf;                          // [D] gated wkv * r_att (input vector of k_attout)
    double *partS;                        // [gridDim.x] partial sums of gated wkv * o_att
    const Ctl *ctl;
    int D;
};

// ln1 -> mix -> K,V,R dequant-GEMV -> WKV (rwkv.cu:535-545; kernels :351-392, :58-100, :221-259)
template <int S>
__global__ __launch_bounds__(NT) void k_att(AttArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XV = S * 1024;
    constexpr int E = XV / NT;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    float *stash = xv + 3 * XV;
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;

    // (1) prologue inputs
    double xl[E], pv[E];
    f32x4 P[E][3];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT, jj = j < D ? j : D - 1;
        xl[e] = a.x[jj];
        pv[e] = a.sxy[so + jj];
#pragma unroll
        for (int q = 0; q < 3; q++) P[e][q] = a.pk[jj * 3 + q];
    }
    // (2) step 0 of the first group's weights.  Only one step: a wave that asks for more than the
    // memory pipe accepts stalls AT ISSUE (in-order), and the prologue below would wait with it.
    // Unconditional (a wave without a group re-reads a neighbour's rows): a branch around the loads
    // would make hipcc's waitcnt pass assume the no-load path and drain the weights early.
    u32x4 w[3][S];
    int g = g0 + wave;
    const uint8_t *wb = a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 3 * D;
    group_load<3, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);

    // (3) LayerNorm, token-shift mix, pre-scale by the per-row quantisation scale, stage in LDS
    double mean, rstd;
    ln_stats<E>(xl, D, mean, rstd, red);
    double Ssum[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT;
        float xk = 0.f, xvv = 0.f, xr = 0.f;
        if (j < D) {
            const double xx = (double)P[e][0][0] * ((xl[e] - mean) * rstd) + (double)P[e][0][1];
            const double prev = pv[e];
            const double mk = (double)P[e][0][2], mv = (double)P[e][0][3], mr = (double)P[e][1][0];
            const float fk = (float)(mk * xx + (1.0 - mk) * prev);   // rwkv.cu:382-384: rounded to f32
            const float fv = (float)(mv * xx + (1.0 - mv) * prev);
            const float fr = (float)(mr * xx + (1.0 - mr) * prev);
            xk = fk * P[e][1][1]; xvv = fv * P[e][1][2]; xr = fr * P[e][1][3];
            Ssum[0] += (double)(fk * P[e][2][0]); Ssum[1] += (double)(fv * P[e][2][1]); Ssum[2] += (double)(fr * P[e][2][2]);
            if (blockIdx.x == 0) a.xx_buf[j] = xx;
        }
        const int p = xperm(j);
        xv[p] = xk; xv[XV + p] = xvv; xv[2 * XV + p] = xr;
    }
    group_load<3, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);   // remaining steps of the first group
    __syncthreads();                                                       // staged vectors visible

    for (; g < g1; g += NW) {
        float out[3];
        const bool nv = g + NW < g1;
        group_dot<3, S, PAT_PER_ROW>(w, xv, XV, lane, out, a.w + (size_t)(nv ? g + NW : 0) * 3 * D, (size_t)D, chunks, nv);
        if (lane == 0) { stash[(g - g0) * 3 + 0] = out[0]; stash[(g - g0) * 3 + 1] = out[1]; stash[(g - g0) * 3 + 2] = out[2]; }
    }
    block_sum<3>(Ssum, red);   // offset terms: only needed by the epilogue (also the barrier before it)

    // WKV recurrence + receptance gate, one lane per channel (rwkv.cu:242-255)
    double part[1] = {0.0};
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
        a.ybuf[i] = yf * a.r_att[i];
        part[0] = (double)(yf * a.o_att[i]);
    }
    block_sum<1>(part, red);
    if (threadIdx.x == 0) a.partS[blockIdx.x] = part[0];
}

// ------------------------------------------------------------------------------------------
struct AttOutArgs {
    const uint8_t *w;      // [D][D] u8 rows = output channels
    const float *ybuf;     // [D] pre-scaled input vector
    const double *partS;   // [n_part] partial offset sums (n_part <= NT)
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
    constexpr int XV = S * 1024;
    constexpr int E = XV / NT;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    const int G = (D + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    float yl[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int j = threadIdx.x + e * NT; yl[e] = a.ybuf[j < D ? j : D - 1]; }
    double Ssum[1] = {a.partS[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};
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
#pragma unroll
    for (int e = 0; e < E; e++) { const int j = threadIdx.x + e * NT; xv[xperm(j)] = (j < D) ? yl[e] : 0.f; }
    if ((int)threadIdx.x >= a.n_part) Ssum[0] = 0.0;
    group_load<R, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);
    block_sum<1>(Ssum, red);
    const float Sf = (float)Ssum[0];

    for (; g < g1; g += NW) {
        float out[R];
        const bool nv = g + NW < g1;
        group_dot<R, S, PAT_SHARED>(w, xv, XV, lane, out, rowbase(nv ? g + NW : 0), (size_t)D, chunks, nv);
        int row0 = g * R;
        const int shift = (row0 > D - R) ? row0 - (D - R) : 0;   // last group may overlap the previous one
        row0 -= shift;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (lane == r && r >= shift) {
                const int i = row0 + r;
                const float acc = (float)a.x[i] + (out[r] + Sf);   // f32 accumulator pre-loaded with x (:548)
                a.x[i] = (double)acc;                                // :553
                a.sxy[so + i] = a.xx_buf[i];                         // mixatt's state write (:385), deferred
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
struct FfnRKArgs {
    const double *x;
    const f32x4 *pk;                  // [D][2] packed {lnw,lnb,mixk,mixr | rk,ok,rr,or}
    const uint8_t *w;                 // [D][5][D]: rows ffn_k out 4i..4i+3, then ffn_r out i
    const float *r_fv, *o_fv;         // ffn_v scale / offset [4D]
    const double *sdd;                // state dd of this layer (read only here)
    size_t slot_stride;
    double *xx_buf;                   // [D] ln2 output -> committed to sdd by k_ffnv
    float *hbuf;                      // [4D] relu^2(k) * r_fv
    float *rgate;                     // [D] sigmoid(r)
    double *partS;                    // [gridDim.x]
    const Ctl *ctl;
    int D;
    unsigned long long *tl;           // optional phase timeline (see tl_stamp)
};

// ln2 -> mix -> ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2 (rwkv.cu:557-573)
template <int S>
__global__ __launch_bounds__(NT) void k_ffn_rk(FfnRKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XV = S * 1024;
    constexpr int E = XV / NT;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    float *stash = xv + 2 * XV;
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;
    tl_stamp(a.tl, 0);

    double xl[E], pv[E];
    f32x4 P[E][2];
#ifdef RWKV_EXP_NOPROLOGUE   // timing experiment only: no prologue traffic, constant staged vectors
#pragma unroll
    for (int e = 0; e < E; e++) { xl[e] = 1.0 + e; pv[e] = 0.5; P[e][0] = f32x4{1.f, 0.f, 0.5f, 0.5f}; P[e][1] = f32x4{0.01f, 0.f, 0.01f, 0.f}; }
#else
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT, jj = j < D ? j : D - 1;
        xl[e] = a.x[jj];
        pv[e] = a.sdd[so + jj];
        P[e][0] = a.pk[jj * 2];
        P[e][1] = a.pk[jj * 2 + 1];
    }
#endif
    u32x4 w[5][S];
    int g = g0 + wave;
#ifdef RWKV_EXP_SAMEROWS   // timing experiment only: every wave streams group 0 (cache resident) -> compute-only time
#define RWKV_EXP_G(gg) 0
#else
#define RWKV_EXP_G(gg) (gg)
#endif
    const uint8_t *wb = a.w + (size_t)RWKV_EXP_G(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 5 * D;
    group_load<5, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);
    tl_stamp(a.tl, 1);

    double mean, rstd;
#ifdef RWKV_EXP_NOPROLOGUE
    mean = 0.25; rstd = 0.9;
#else
    ln_stats<E>(xl, D, mean, rstd, red);
#endif
    tl_stamp(a.tl, 2);
    double Ssum[2] = {0.0, 0.0};
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT;
        float xk = 0.f, xr = 0.f;
        if (j < D) {
            const double xx = (double)P[e][0][0] * ((xl[e] - mean) * rstd) + (double)P[e][0][1];
            const double prev = pv[e];
            const double mk = (double)P[e][0][2], mr = (double)P[e][0][3];
            const float fk = (float)(mk * xx + (1.0 - mk) * prev);   // f64 mix (:341-342), f32 cast in the GEMV (:290)
            const float fr = (float)(mr * xx + (1.0 - mr) * prev);
            xk = fk * P[e][1][0]; xr = fr * P[e][1][2];
            Ssum[0] += (double)(fk * P[e][1][1]); Ssum[1] += (double)(fr * P[e][1][3]);
            if (blockIdx.x == 0) a.xx_buf[j] = xx;
        }
        const int p = xperm(j);
        xv[p] = xk; xv[XV + p] = xr;
    }
    group_load<5, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);
    __syncthreads();
    tl_stamp(a.tl, 3);
    bool first_ = true;

    for (; g < g1; g += NW) {
        float out[5];
        const bool nv = g + NW < g1;
        group_dot<5, S, PAT_FFN_RK>(w, xv, XV, lane, out, a.w + (size_t)RWKV_EXP_G(nv ? g + NW : 0) * 5 * D, (size_t)D, chunks, nv);
#pragma unroll
        for (int r = 0; r < 5; r++)
            if (lane == r) stash[(g - g0) * 5 + r] = out[r];
        if (first_) { tl_stamp(a.tl, 4); first_ = false; }
    }
    block_sum<2>(Ssum, red);   // offset terms: only needed by the epilogue (also the barrier before it)
    tl_stamp(a.tl, 5);

    double part[1] = {0.0};
    for (int t = threadIdx.x; t < 5 * (g1 - g0); t += NT) {
        const int q = t % 5, i = g0 + t / 5;
        const float val = stash[t] + (float)Ssum[q < 4 ? 0 : 1];
        if (q < 4) {
            float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
            h = h * h;
            const int kk = 4 * i + q;
            a.hbuf[kk] = h * a.r_fv[kk];
            part[0] += (double)(h * a.o_fv[kk]);
        } else {
            a.rgate[i] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
        }
    }
    block_sum<1>(part, red);
    if (threadIdx.x == 0) a.partS[blockIdx.x] = part[0];
    tl_stamp(a.tl, 6);
}

// ------------------------------------------------------------------------------------------
struct FfnVArgs {
    const uint8_t *w;      // [D][4D] u8: row i = output channel i, as 4 quarter-rows of D bytes
    const float *hbuf;     // [4D] pre-scaled hidden vector
    const double *partS;
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
    constexpr int XV = S * 1024;
    constexpr int E = XV / NT;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    const int g0 = (int)(((long long)blockIdx.x * D) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);

    float hl[4][E];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < E; e++) { const int j = threadIdx.x + e * NT; hl[q][e] = a.hbuf[q * D + (j < D ? j : D - 1)]; }
    double Ssum[1] = {a.partS[(int)threadIdx.x < a.n_part ? threadIdx.x : 0]};
    const size_t so = (size_t)a.ctl->slot * a.slot_stride;

    u32x4 w[4][S];
    int g = g0 + wave;
    const uint8_t *wb = a.w + (size_t)(g < g1 ? g : (g1 > g0 ? g1 - 1 : 0)) * 4 * D;
    group_load<4, S, 0, pre_steps<S>()>(w, wb, (size_t)D, chunks, lane);

#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < E; e++) { const int j = threadIdx.x + e * NT; xv[q * XV + xperm(j)] = (j < D) ? hl[q][e] : 0.f; }
    if ((int)threadIdx.x >= a.n_part) Ssum[0] = 0.0;
    group_load<4, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);
    block_sum<1>(Ssum, red);
    const float Sf = (float)Ssum[0];

    for (; g < g1; g += NW) {
        float out[4];
        const bool nv = g + NW < g1;
        group_dot<4, S, PAT_PER_ROW>(w, xv, XV, lane, out, a.w + (size_t)(nv ? g + NW : 0) * 4 * D, (size_t)D, chunks, nv);
        if (lane == 0) {
            const float v = ((out[0] + out[1]) + (out[2] + out[3])) + Sf;
            a.x[g] = a.x[g] + (double)(v * a.rgate[g]);   // blockout, rwkv.cu:407 (f32 product)
            a.sdd[so + g] = a.xx_buf[g];                   // mixffn's state write (:344), deferred
        }
    }
}

// ------------------------------------------------------------------------------------------
struct HeadArgs {
    const double *x;
    const f32x4 *pk;           // [D] packed {lnw, lnb, r, o} of ln_out / head
    const uint8_t *w;          // [V][D] u8
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
    constexpr int XV = S * 1024;
    constexpr int E = XV / NT;
    constexpr int R = 4;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    float *bval = xv + XV;
    unsigned *bidx = reinterpret_cast<unsigned *>(bval + NW);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    const int V = (int)VOCAB;
    const int G = (V + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    double xl[E];
    f32x4 P[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT, jj = j < D ? j : D - 1;
        xl[e] = a.x[jj];
        P[e] = a.pk[jj];
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
    ln_stats<E>(xl, D, mean, rstd, red);
    double Ssum[1] = {0.0};
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int j = threadIdx.x + e * NT;
        float xs = 0.f;
        if (j < D) {
            const float f = (float)((double)P[e][0] * ((xl[e] - mean) * rstd) + (double)P[e][1]);
            xs = f * P[e][2];
            Ssum[0] += (double)(f * P[e][3]);
        }
        xv[xperm(j)] = xs;
    }
    group_load<R, S, pre_steps<S>(), S>(w, wb, (size_t)D, chunks, lane);
    block_sum<1>(Ssum, red);
    const float Sf = (float)Ssum[0];
    float *lg = a.logits + (size_t)a.ctl->out_row * V;

    float best = -INFINITY;
    unsigned besti = 0xffffffffu;
    for (; g < g1; g += NW) {
        float out[R];
        const bool nv = g + NW < g1;
        group_dot<R, S, PAT_SHARED>(w, xv, XV, lane, out, rowbase(nv ? g + NW : 0), (size_t)D, chunks, nv);
        int row0 = g * R;
        const int shift = (row0 > V - R) ? row0 - (V - R) : 0;
        row0 -= shift;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float val = out[r] + Sf;
            const int i = row0 + r;
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
// cudac_mm8_one, rwkv.cu:267-311).  w_t is the re-tiled [M][N] matrix.  N <= 5120: rows whole;
// otherwise N = 4*Dq and rows are processed as 4 quarter-rows (the ffn_v shape).
struct Mm8Args {
    const uint8_t *w_t;
    const float *x, *r, *o;
    float *y;
    int N, M;
};
template <int S, bool QUARTERS>
__global__ __launch_bounds__(NT) void k_mm8(Mm8Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XV = S * 1024;
    constexpr int R = 4;
    double *red = reinterpret_cast<double *>(smem);
    float *xv = reinterpret_cast<float *>(smem + RED_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Dq = QUARTERS ? a.N / 4 : a.N;
    const int chunks = Dq >> 4;
    const int M = a.M;
    const int G = QUARTERS ? M : (M + R - 1) / R;
    const int g0 = (int)(((long long)blockIdx.x * G) / gridDim.x);
    const int g1 = (int)(((long long)(blockIdx.x + 1) * G) / gridDim.x);

    double Ssum[1] = {0.0};
    for (int q = 0; q < (QUARTERS ? 4 : 1); q++)
        for (int j = threadIdx.x; j < XV; j += NT) {
            float xs = 0.f;
            if (j < Dq) {
                const float f = a.x[q * Dq + j];
                xs = f * a.r[q * Dq + j];
                Ssum[0] += (double)(f * a.o[q * Dq + j]);
            }
            xv[q * XV + xperm(j)] = xs;
        }
    block_sum<1>(Ssum, red);
    const float Sf = (float)Ssum[0];

    for (int g = g0 + wave; g < g1; g += NW) {
        u32x4 w[R][S];
        float out[R];
        if (QUARTERS) {
            group_load<R, S, 0, S>(w, a.w_t + (size_t)g * a.N, (size_t)Dq, chunks, lane);
            group_dot<R, S, PAT_PER_ROW>(w, xv, XV, lane, out, a.w_t, 0, chunks, false);
            if (lane == 0) a.y[g] = ((out[0] + out[1]) + (out[2] + out[3])) + Sf;
        } else {
            int row0 = g * R;
            const int shift = (row0 > M - R) ? row0 - (M - R) : 0;
            row0 -= shift;
            group_load<R, S, 0, S>(w, a.w_t + (size_t)row0 * a.N, (size_t)a.N, chunks, lane);
            group_dot<R, S, PAT_SHARED>(w, xv, XV, lane, out, a.w_t, 0, chunks, false);
#pragma unroll
            for (int r = 0; r < R; r++)
                if (lane == r && r >= shift) a.y[row0 + r] = out[r] + Sf;
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

// uw = bonus + decay, ew = exp(decay)   (constants of rwkv.cu:247-252, hoisted out of the token loop)
__global__ void k_prep_wkv(const double *decay, const double *bonus, double *uw, double *ew, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uw[i] = bonus[i] + decay[i]; ew[i] = exp(decay[i]); }
}

// Load-time packing of the static per-channel parameters into float4 tables (the prologues read
// 1-3 16-byte pieces per element instead of up to 13 scalars).  The f64 tensors involved
// (layernorm rows, time-mix vectors) hold f32-representable values in converted checkpoints
// (converter: .double() of f32 tensors, convert_model.py:44-56), so the narrowing is exact there.
__global__ void k_pack_att(f32x4 *pk, const double *lnw, const double *lnb, const double *mk, const double *mv,
                           const double *mr, const float *rk, const float *rv, const float *rr,
                           const float *ok, const float *ov, const float *orr, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    pk[j * 3 + 0] = f32x4{(float)lnw[j], (float)lnb[j], (float)mk[j], (float)mv[j]};
    pk[j * 3 + 1] = f32x4{(float)mr[j], rk[j], rv[j], rr[j]};
    pk[j * 3 + 2] = f32x4{ok[j], ov[j], orr[j], 0.f};
}
__global__ void k_pack_ffn(f32x4 *pk, const double *lnw, const double *lnb, const double *mk, const double *mr,
                           const float *rk, const float *ok, const float *rr, const float *orr, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    pk[j * 2 + 0] = f32x4{(float)lnw[j], (float)lnb[j], (float)mk[j], (float)mr[j]};
    pk[j * 2 + 1] = f32x4{rk[j], ok[j], rr[j], orr[j]};
}
__global__ void k_pack_head(f32x4 *pk, const double *lnw, const double *lnb, const float *r, const float *o, int D)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    pk[j] = f32x4{(float)lnw[j], (float)lnb[j], r[j], o[j]};
}

} // namespace rwkvk
