// seq.hip.h -- gfx950 device code of the CHUNKED (prompt prefill) forward: up to 32 tokens of one
// sequence per pass, every uint8 weight byte read ONCE per chunk instead of once per token.
//
// Replaces, for MODE GPT with several tokens, the reference's per-op loops over `tokenlength`
// (rwkv.cu:493-593 with tokenlength > 1; mm8 kernels :58-142,:267-311 re-stream the weights per
// token), i.e. what RWKV::loadContext drives (rwkv.h:395-413).  Same arithmetic contract as the
// single-token kernels (kernels.hip.h): activations are quantised per (vector, token) to 23-bit
// fixed point, split into three byte limbs, and contracted against the uint8 weights in exact
// integer arithmetic -- here on the matrix cores: v_mfma_i32_16x16x64_i8 (a real int8 contraction:
// M = 16 tokens, N = 16 weight rows, K = 64).  Both operands are stored signed (byte - 128); the
// offsets are removed with the weight row sums and the per-token limb sums:
//     sum_k u_k l_k = MFMA(u - 128, l - 128) + 128 * rowsum(u) + 128 * sum_k (l_k - 128).
//
//   k_seq_embed   rwkv.cu:513-524   embedding rows + ln0 for the chunk
//   k_seq_site    :412-465,:313-392 LayerNorm + token-shift mix (the shift runs along the chunk; token 0
//                                   takes the recurrent state) + quantisation into the MFMA A-operand image
//   k_mm8_seq     :58-142,:267-311  [T x K] x [K x N] uint8 GEMM on MFMA, one launch per matrix group;
//                 :548-553,:574-577 the residual updates are its epilogues
//   k_seq_wkv     :221-259          WKV recurrence, sequential over the chunk per channel
//   k_seq_stage   :144-219          relu^2 / cast + scale + quantisation for att_out and ffn_v inputs
#pragma once
#include "kernels.hip.h"

namespace rwkvk {

constexpr int SEQ_T = 32;                 // tokens per pass: two 16-row MFMA tiles
constexpr double SEQ_CU = 4227200.0;      // 128 * (1 + 256 + 65536) - 2^22: weight-row-sum coefficient
typedef int i32x4 __attribute__((ext_vector_type(4)));

// A-operand image of one activation vector for a chunk: 16-byte units indexed
// [k-block of 64][token tile of 16][limb][lane], lane = 16 * ((k % 64) / 16) + token % 16 -- exactly the
// register image of v_mfma_i32_16x16x64_i8's A operand, so a wave loads a fragment with one
// coalesced 1 KiB global_load_dwordx4.
__device__ __forceinline__ size_t a_unit(int kb, int mt, int limb, int lane) { return (((size_t)kb * 2 + mt) * 3 + limb) * 64 + lane; }
__host__ __device__ inline size_t a_image_bytes(size_t K) { return (K / 64) * 2 * 3 * 64 * 16; }

// per (vector, token) quantisation record
struct SeqVec {
    double scale;   // real value of one quantisation step
    double cA;      // 128 * sum_b 256^b * sum_k (limb_b[k] - 128)
    float So;       // offset term sum_k f[k] * o[k]
    float pad;
};

// quantise one quad (4 consecutive k) of token t and store its three signed limb dwords; returns the
// sums of the unsigned limb bytes in s[0..2]
__device__ __forceinline__ void seq_store_quad(unsigned *img, int qd, int t, const float (&xr)[4], float inv_s, unsigned (&s)[3])
{
    unsigned q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) q[e] = __float_as_uint(fmaf(xr[e], inv_s, QMAGIC));
    const unsigned p01 = __builtin_amdgcn_perm(q[1], q[0], 0x05010400u), p23 = __builtin_amdgcn_perm(q[3], q[2], 0x05010400u);
    const unsigned h01 = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0602u), h23 = __builtin_amdgcn_perm(q[3], q[2], 0x0c0c0602u);
    const unsigned d[3] = {__builtin_amdgcn_perm(p23, p01, 0x05040100u), __builtin_amdgcn_perm(p23, p01, 0x07060302u),
                           __builtin_amdgcn_perm(h23, h01, 0x05040100u)};
    const int c = qd >> 2, dq = qd & 3;                 // 16-byte unit along k, dword inside it
    const int kb = c >> 2, lane = 16 * (c & 3) + (t & 15), mt = t >> 4;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        img[a_unit(kb, mt, b, lane) * 4 + dq] = d[b] ^ 0x80808080u;
        s[b] += __builtin_amdgcn_udot4(d[b], 0x01010101u, 0u, false);
    }
}

// workgroup-wide reduction of 3*NV limb sums, then thread 0 writes the records of token t
template <int NV>
__device__ __forceinline__ void seq_finish(unsigned (&ls)[NV][3], const double (&So)[NV], const float (&amax)[NV], int K, int t, SeqVec *rec,
                                           unsigned *lds_sums)
{
    if (threadIdx.x < 3 * NV) lds_sums[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NV; m++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const unsigned v = wave_sum_dpp(ls[m][b]);
            if ((threadIdx.x & 63) == 0) atomicAdd(&lds_sums[m * 3 + b], v);
        }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int m = threadIdx.x;
        double ca = 0.0, f = 128.0;
        for (int b = 0; b < 3; b++) { ca += f * ((double)lds_sums[m * 3 + b] - 128.0 * (double)K); f *= 256.0; }
        SeqVec r;
        r.scale = scale_of(amax[m]); r.cA = ca; r.So = (float)So[m]; r.pad = 0.f;
        rec[m * SEQ_T + t] = r;
    }
}

// real value of one GEMM output from the raw integer contraction (kept for documentation; the GEMM
// kernel applies it inline): scale * (M0 + 256 M1 + 65536 M2 + cA + SEQ_CU * rowsum) + So

// ------------------------------------------------------------------------------------------
struct SeqEmbedArgs {
    const float *embed;
    const double *ln;            // rows 0,1 = ln0
    const unsigned long long *tokens;
    double *x;                   // [T][D]
    int D;
};
__global__ __launch_bounds__(NT) void k_seq_embed(SeqEmbedArgs a)
{
    __shared__ double red[RED_BYTES / 8];
    const int D = a.D, t = blockIdx.x;
    const float *row = a.embed + (size_t)a.tokens[t] * D;
    double s[2] = {0.0, 0.0};
    for (int j = threadIdx.x; j < D; j += NT) { const double v = (double)row[j]; s[0] += v; s[1] += v * v; }
    block_sum<2>(s, red + RED_STATS);
    const double mean = s[0] / (double)D, rstd = 1.0 / sqrt((s[1] - s[0] * mean) / (double)(D - 1));
    for (int j = threadIdx.x; j < D; j += NT) a.x[(size_t)t * D + j] = a.ln[j] * (((double)row[j] - mean) * rstd) + a.ln[D + j];
}

// ------------------------------------------------------------------------------------------
struct SeqSiteArgs {
    const double *x;             // [T][D] residual stream (the GEMM epilogues keep it up to date)
    const double *lnw, *lnb;     // this site's LayerNorm rows
    const double *mix[3];        // token-shift mix per vector (nullptr: no shift, ln_out -> head)
    const float *r[3], *o[3];    // scale / offset of the matrices the vectors feed
    const double *state;         // previous LayerNorm output (state xy / dd of this layer, slot 0): token 0's shift input
    double *state_new;           // GPT: [D] LayerNorm output of the last token (copied over the state afterwards)
    int par;                     // PARRALEL mode (rwkv.cu:236-240): row t is an independent sequence with state slot slot0 + t --
    double *state_par;           //   shift input and state write both go to that slot of this array (same base as `state`)
    size_t slot_stride;          //   L * D
    int slot0;
    unsigned *img[3];            // A-operand images
    SeqVec *rec;                 // [NV][SEQ_T]
    int D, T;
};

constexpr int SEQ_SNT = 512;     // threads of a site workgroup (one workgroup per row)
constexpr int SEQ_SNW = SEQ_SNT / 64;

// workgroup reductions for SEQ_SNT threads (kernels.hip.h's block_sum / block_max are sized for NT)
template <int K>
__device__ __forceinline__ void sblock_sum(double (&v)[K], double *red)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[w * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < SEQ_SNW; i++) t += red[i * K + k];
        v[k] = t;
    }
}
template <int K>
__device__ __forceinline__ void sblock_max(float (&v)[K], double *redd)
{
    float *red = reinterpret_cast<float *>(redd);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_max(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[w * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < SEQ_SNW; i++) t = fmaxf(t, red[i * K + k]);
        v[k] = t;
    }
}

// NQ quads per thread: 2 for D <= 4096, 3 up to 6144.  Every input -- the two rows of x and all parameter
// vectors -- is requested with 16-byte loads before the first reduction: one workgroup per row is latency
// bound, and element-wise scalar loads of the ~11 parameter vectors cost 10 of its 22 us.
template <int NV, int NQ>
__global__ __launch_bounds__(SEQ_SNT) void k_seq_site(SeqSiteArgs a)
{
    __shared__ double red[SEQ_SNW * 4];
    __shared__ unsigned lds_sums[12];
    const int D = a.D, t = blockIdx.x, nqd = D >> 2;
    const bool shift = a.mix[0] != nullptr;
    const bool lnprev = shift && t > 0 && !a.par;   // the shift input is a LayerNorm output: of the previous row (GPT) or already stored (state)
    const double *xprow = !shift ? a.x : a.par ? a.state_par + (size_t)(a.slot0 + t) * a.slot_stride : (t > 0 ? a.x + (size_t)(t - 1) * D : a.state);
    double xt[NQ][4], xp[NQ][4], lw[NQ][4], lb[NQ][4], mk[NQ][NV][4];
    f32x4 rr[NQ][NV], oo[NQ][NV];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * SEQ_SNT, qc = qd < nqd ? qd : nqd - 1;
        load_quad_f64(a.x + (size_t)t * D, qc, xt[i]);
        load_quad_f64(xprow, qc, xp[i]);
        load_quad_f64(a.lnw, qc, lw[i]);
        load_quad_f64(a.lnb, qc, lb[i]);
#pragma unroll
        for (int m = 0; m < NV; m++) {
            if (shift) load_quad_f64(a.mix[m], qc, mk[i][m]);
            rr[i][m] = reinterpret_cast<const f32x4 *>(a.r[m])[qc];
            oo[i][m] = reinterpret_cast<const f32x4 *>(a.o[m])[qc];
        }
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NQ; i++)
        if ((int)(threadIdx.x + i * SEQ_SNT) < nqd)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s[0] += xt[i][e]; s[1] += xt[i][e] * xt[i][e];
                if (lnprev) { s[2] += xp[i][e]; s[3] += xp[i][e] * xp[i][e]; }
            }
    sblock_sum<4>(s, red);
    const double mean = s[0] / (double)D, rstd = 1.0 / sqrt((s[1] - s[0] * mean) / (double)(D - 1));
    const double meanp = s[2] / (double)D, rstdp = lnprev ? 1.0 / sqrt((s[3] - s[2] * meanp) / (double)(D - 1)) : 1.0;

    float xr[NV][NQ][4];
    double So[NV];
    float amax[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { So[m] = 0.0; amax[m] = 0.f; }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * SEQ_SNT;
        const bool real = qd < nqd;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = qd * 4 + e;
            const double xx = lw[i][e] * ((xt[i][e] - mean) * rstd) + lb[i][e];
            double xprev = xp[i][e];
            if (lnprev) xprev = lw[i][e] * ((xp[i][e] - meanp) * rstdp) + lb[i][e];
            if (real && a.par) a.state_par[(size_t)(a.slot0 + t) * a.slot_stride + j] = xx;   // own slot: read above by this thread only
            else if (real && a.state_new && t == a.T - 1) a.state_new[j] = xx;          // mixatt / mixffn state write (:344,:385)
#pragma unroll
            for (int m = 0; m < NV; m++) {
                float f = (float)xx;
                if (shift) f = (float)(xx * mk[i][m][e] + xprev * (1.0 - mk[i][m][e]));   // :339-343,:377-384
                const float v = real ? f * rr[i][m][e] : 0.f;
                xr[m][i][e] = v;
                if (real) { So[m] += (double)(f * oo[i][m][e]); amax[m] = fmaxf(amax[m], fabsf(v)); }
            }
        }
    }
    sblock_sum<NV>(So, red);
    sblock_max<NV>(amax, red);
    unsigned ls[NV][3];
#pragma unroll
    for (int m = 0; m < NV; m++) {
        ls[m][0] = ls[m][1] = ls[m][2] = 0u;
        const float inv_s = inv_scale(amax[m]);
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int qd = threadIdx.x + i * SEQ_SNT;
            if (qd < nqd) seq_store_quad(a.img[m], qd, t, xr[m][i], inv_s, ls[m]);
        }
    }
    seq_finish<NV>(ls, So, amax, D, t, a.rec, lds_sums);
}

// ------------------------------------------------------------------------------------------
struct SeqStageArgs {
    const float *src;            // kind 0: gated wkv y [T][D]; kind 1: ffn k/r GEMM output [T][5D], element [t][5 i + q]
    const float *r, *o;          // scale / offset over K
    unsigned *img;
    SeqVec *rec;                 // [SEQ_T]
    int K, T;
};
constexpr int SEQ_NQS = 10;      // K <= 4 * 5120 -> 5120 quads over 512 threads

// KIND 0: f = src (att_out input).  KIND 1: f = relu(k)^2 with k = ffn_k GEMM output (rwkv.cu:189-190), K = 4D
template <int KIND>
__global__ __launch_bounds__(NT) void k_seq_stage(SeqStageArgs a)
{
    __shared__ double red[RED_BYTES / 8];
    __shared__ unsigned lds_sums[12];
    const int K = a.K, t = blockIdx.x, nqd = K >> 2;
    // all inputs of the thread's quads are requested first, 16 bytes at a time (one workgroup per row is latency bound)
    f32x4 sv[SEQ_NQS], rv[SEQ_NQS], ov[SEQ_NQS];
#pragma unroll
    for (int i = 0; i < SEQ_NQS; i++) {
        const int qd = threadIdx.x + i * NT, qc = qd < nqd ? qd : nqd - 1;
        if (KIND == 0) sv[i] = reinterpret_cast<const f32x4 *>(a.src + (size_t)t * K)[qc];
        else __builtin_memcpy(&sv[i], a.src + (size_t)t * (K / 4 * 5) + (size_t)qc * 5, 16);   // k0..k3 of channel qc: 4-byte aligned only
        rv[i] = reinterpret_cast<const f32x4 *>(a.r)[qc];
        ov[i] = reinterpret_cast<const f32x4 *>(a.o)[qc];
    }
    float xr[SEQ_NQS][4];
    double So[1] = {0.0};
    float amax[1] = {0.f};
#pragma unroll
    for (int i = 0; i < SEQ_NQS; i++) {
        const int qd = threadIdx.x + i * NT;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xr[i][e] = 0.f;
            if (qd < nqd) {
                float f = sv[i][e];
                if (KIND == 1) { f = f * (float)(f > 0.f); f = f * f; }     // relu(k)^2, rwkv.cu:189-190
                xr[i][e] = f * rv[i][e];
                So[0] += (double)(f * ov[i][e]);
                amax[0] = fmaxf(amax[0], fabsf(xr[i][e]));
            }
        }
    }
    block_sum<1>(So, red + RED_OFFS);
    block_max<1>(amax, red + RED_MAX);
    unsigned ls[1][3] = {{0u, 0u, 0u}};
    const float inv_s = inv_scale(amax[0]);
#pragma unroll
    for (int i = 0; i < SEQ_NQS; i++) {
        const int qd = threadIdx.x + i * NT;
        if (qd < nqd) seq_store_quad(a.img, qd, t, xr[i], inv_s, ls[0]);
    }
    seq_finish<1>(ls, So, amax, K, t, a.rec, lds_sums);
}

// ------------------------------------------------------------------------------------------
struct SeqWkvArgs {
    const float *kvr;            // K/V/R GEMM output [T][3D], element [t][3 i + m]
    const double *uw, *ew;       // bonus + decay, exp(decay) of this layer
    double *saa, *sbb;           // state of this layer, slot 0
    float *y;                    // [T][D] gated wkv, cast to f32 as the att_out GEMV does (rwkv.cu:290)
    int D, T;
    int par;                     // PARRALEL mode: row t uses state slot slot0 + t (no recurrence along the rows)
    size_t slot_stride;
    int slot0;
};
constexpr int WKV_CH = 8;        // channels per workgroup (256 threads = 8 channels x 32 tokens)
// rwkv.cu:242-255 with the GPT-mode state slot 0.  The exponentials do not depend on the state, so
// one thread per (token, channel) evaluates them; then one thread per channel runs the recurrence
// along the chunk (a division and a few fma per step).
__global__ __launch_bounds__(256) void k_seq_wkv(SeqWkvArgs a)
{
    __shared__ double e1s[SEQ_T][WKV_CH], eks[SEQ_T][WKV_CH], vs[SEQ_T][WKV_CH], sgs[SEQ_T][WKV_CH];
    const int ch = threadIdx.x & (WKV_CH - 1), t = threadIdx.x / WKV_CH;
    const int i = blockIdx.x * WKV_CH + ch;
    if (i < a.D && t < a.T) {
        const float *p = a.kvr + (size_t)t * 3 * a.D + 3 * i;
        const float k = p[0], v = p[1], r = p[2];
        e1s[t][ch] = exp(a.uw[i] + (double)k);
        eks[t][ch] = exp((double)k);
        vs[t][ch] = (double)v;
        sgs[t][ch] = 1.0 / (1.0 + (double)expf(-r));       // rwkv.cu:250: exp of a float argument
    }
    __syncthreads();
    if (a.par) {
        if (i < a.D && t < a.T) {
            const size_t so = (size_t)(a.slot0 + t) * a.slot_stride + i;
            const double aa = a.saa[so], bb = a.sbb[so], ew = a.ew[i];
            const double e1 = e1s[t][ch], ek = eks[t][ch], vv = vs[t][ch];
            a.y[(size_t)t * a.D + i] = (float)(sgs[t][ch] * ((aa + e1 * vv) / (bb + e1)));
            a.saa[so] = (aa + ek * vv) * ew;
            a.sbb[so] = (bb + ek) * ew;
        }
        return;
    }
    if (threadIdx.x < WKV_CH && i < a.D) {
        double aa = a.saa[i], bb = a.sbb[i];
        const double ew = a.ew[i];
        for (int tt = 0; tt < a.T; tt++) {
            const double e1 = e1s[tt][ch], ek = eks[tt][ch], vv = vs[tt][ch];
            const double y = sgs[tt][ch] * ((aa + e1 * vv) / (bb + e1));
            aa = (aa + ek * vv) * ew;
            bb = (bb + ek) * ew;
            a.y[(size_t)tt * a.D + i] = (float)y;
        }
        a.saa[i] = aa; a.sbb[i] = bb;
    }
}

// ------------------------------------------------------------------------------------------
struct SeqGemmArgs {
    const uint8_t *w;            // re-tiled weights, row-per-output: [N][K]
    const unsigned *rs;          // [N] row sums
    int N, K;
    int Q;                       // row classes interleaved in w: row = Q * channel + q (K/V/R: 3; ffn k,k,k,k,r: 5; else 1)
    int vec_of_q[5];             // activation vector each class multiplies
    const u32x4 *img[3];         // A-operand images of the vectors
    const SeqVec *rec;           // [NV][SEQ_T]
    float *out;                  // epi 0: [T][N] f32
    int epi;                     // 0 store; 1 x = f32(x) + v (att_out residual, rwkv.cu:548-553); 2 x += v * sigmoid(gate) (:574-577,:407,:212)
    double *x;                   // epi 1, 2: residual stream [T][N], updated in place (one owner per element)
    const float *gate;           // epi 2: ffn k/r GEMM output [T][5N], the r value of channel j at [t][5 j + 4]
    int T;
    const double *cp_src;        // piggy-back copy (stream-ordered behind the site kernel that produced it): the chunk's
    double *cp_dst;              // last LayerNorm output -> recurrent state; cp_n == 0: none
    int cp_n;
};
constexpr int SEQ_TB = 5;        // weight tiles (16 rows each) per workgroup pass
#ifndef RWKV_SEQ_NT
#define RWKV_SEQ_NT 256
#endif
#ifndef RWKV_SEQ_NTLOAD
#define RWKV_SEQ_NTLOAD 0
#endif
#ifndef RWKV_SEQ_DEPTH
#define RWKV_SEQ_DEPTH 4
#endif
constexpr int SEQ_NT = RWKV_SEQ_NT;   // GEMM workgroup: 4 waves, one per SIMD, so each may use the full 512-register budget
constexpr int SEQ_NW = SEQ_NT / 64;

// operands of one k-block: B fragments of NTL tiles and the A fragments of the pass's vector
template <int NTL> struct SeqFrag { u32x4 bw[NTL]; u32x4 af[2][3]; };
template <int NTL>
__device__ __forceinline__ void seq_frag_load(SeqFrag<NTL> &f, const uint8_t *const (&wrow)[SEQ_TB], const u32x4 *img, int kb, int lane)
{
#pragma unroll
#ifdef RWKV_SEQ_EXP_NOB      // timing experiment only (wrong results)
    for (int i = 0; i < NTL; i++) f.bw[i] = u32x4{(unsigned)kb, (unsigned)lane, (unsigned)i, 4u};
#elif RWKV_SEQ_NTLOAD
    for (int i = 0; i < NTL; i++) f.bw[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow[i] + (size_t)kb * 64));
#else
    for (int i = 0; i < NTL; i++) f.bw[i] = *reinterpret_cast<const u32x4 *>(wrow[i] + (size_t)kb * 64);
#endif
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
#ifdef RWKV_SEQ_EXP_NOA      // timing experiment only (wrong results)
            f.af[mt][b] = u32x4{(unsigned)kb, (unsigned)lane, 3u, 4u};
#else
            f.af[mt][b] = img[a_unit(kb, mt, b, lane)];
#endif
        }
}
template <int NTL>
__device__ __forceinline__ void seq_frag_mfma(const SeqFrag<NTL> &f, i32x4 (&acc)[SEQ_TB][2][3], int lane)
{
#pragma unroll
    for (int i = 0; i < NTL; i++) {
        // the weights were fetched with lane 4 c + g <- bytes [16 g, +16) of row c (adjacent lanes read adjacent
        // bytes: 64 B per row coalesce); the B operand wants them in lane 16 g + c: one crossbar transposition
        const int src = (4 * (lane & 15) + (lane >> 4)) << 2;
        i32x4 bf;
#pragma unroll
        for (int d = 0; d < 4; d++) bf[d] = __builtin_amdgcn_ds_bpermute(src, (int)(f.bw[i][d] ^ 0x80808080u));   // uint8 weight -> signed operand
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const i32x4 af = i32x4{(int)f.af[mt][b][0], (int)f.af[mt][b][1], (int)f.af[mt][b][2], (int)f.af[mt][b][3]};
                acc[i][mt][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[i][mt][b], 0, 0, 0);
            }
    }
}
// K loop of one pass over NTL tiles.  The waves take k-blocks round-robin (wave w: kb = w, w + 4, ...):
// at any moment the workgroup reads 4 adjacent 64-byte pieces of each weight row, and with the two
// k-blocks requested ahead 768 contiguous bytes per row are in flight -- HBM page locality; a wave
// owning a contiguous K slice instead reads 64-byte pieces 1 KiB apart and ran at a quarter of the rate.
template <int NTL>
__device__ __forceinline__ void seq_pass(i32x4 (&acc)[SEQ_TB][2][3], const uint8_t *const (&wrow)[SEQ_TB], const u32x4 *img, int KB, int wave, int lane)
{
    const int n = (KB - wave + SEQ_NW - 1) / SEQ_NW;      // k-blocks of this wave
    if (n <= 0) return;
    auto kbi = [&](int it) { return wave + SEQ_NW * (it < n - 1 ? it : n - 1); };   // past the end: re-read the last (branch-free loop)
    // ring of DEPTH operand sets: DEPTH - 1 k-blocks are requested ahead of the one being multiplied
    constexpr int DEPTH = RWKV_SEQ_DEPTH;
    SeqFrag<NTL> f[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++) seq_frag_load<NTL>(f[j], wrow, img, kbi(j), lane);
    int it = 0;
    for (; it + DEPTH <= n; it += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            seq_frag_load<NTL>(f[(j + DEPTH - 1) % DEPTH], wrow, img, kbi(it + j + DEPTH - 1), lane);
            seq_frag_mfma<NTL>(f[j], acc, lane);
        }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++)
        if (it + j < n) seq_frag_mfma<NTL>(f[j], acc, lane);
}

// One workgroup owns a contiguous range of (class-major) 16-row tiles and works through it in passes
// of up to SEQ_TB tiles that share an activation vector; its 4 waves split K; operands of the next
// two k-blocks are requested before the MFMAs of the current one issue (120 accumulator registers +
// three operand sets need more than 256 registers, hence one wave per SIMD); the integer partial
// sums meet in LDS (exact: order does not matter), then the workgroup applies scale, offsets and
// corrections.  Weight bytes are read once, 16 B per lane: lane 4 c + g reads bytes [64 kb + 16 g, +16)
// of row c of the tile (coalesced), and a ds_bpermute per dword moves them into the B operand's lane 16 g + c.
#if RWKV_SEQ_NT == 256
__global__ __launch_bounds__(SEQ_NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mm8_seq(SeqGemmArgs a)
#else
__global__ __launch_bounds__(SEQ_NT) void k_mm8_seq(SeqGemmArgs a)
#endif
{
    __shared__ double accl[SEQ_TB][2][4][64];   // [tile][token tile][reg][lane], 20 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nch = (N + Q - 1) / Q;                    // channels per class
    const int CB = (nch + 15) >> 4;                     // 16-channel blocks per class
    const int ntiles = Q * CB;
    const int tb0 = (int)(((long long)blockIdx.x * ntiles) / gridDim.x), tb1 = (int)(((long long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    const int g = lane & 3, c = lane >> 2;   // load mapping: 4 adjacent lanes cover 64 contiguous bytes of a row
    if (blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < a.cp_n; j += SEQ_NT) a.cp_dst[j] = a.cp_src[j];

    int tg = tb0;
    while (tg < tb1) {
        // tiles of this pass: same activation vector
        const int v0 = a.vec_of_q[tg / CB];
        int nt = 1;
        while (nt < SEQ_TB && tg + nt < tb1 && a.vec_of_q[(tg + nt) / CB] == v0) nt++;
        for (int e = threadIdx.x; e < SEQ_TB * 2 * 4 * 64; e += SEQ_NT) (&accl[0][0][0][0])[e] = 0.0;
        __syncthreads();

        const uint8_t *wrow[SEQ_TB];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++) {
            const int id = tg + (i < nt ? i : 0);
            const int q = id / CB, cb = id % CB;
            int row = Q * (16 * cb + c) + q;
            row = row < N ? row : N - 1;
            wrow[i] = a.w + (size_t)row * K + 16 * g;
        }
        const u32x4 *img = a.img[v0];
        i32x4 acc[SEQ_TB][2][3];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int b = 0; b < 3; b++) acc[i][mt][b] = i32x4{0, 0, 0, 0};
        switch (nt) {
        case 1: seq_pass<1>(acc, wrow, img, KB, wave, lane); break;
        case 2: seq_pass<2>(acc, wrow, img, KB, wave, lane); break;
        case 3: seq_pass<3>(acc, wrow, img, KB, wave, lane); break;
        case 4: seq_pass<4>(acc, wrow, img, KB, wave, lane); break;
        default: seq_pass<5>(acc, wrow, img, KB, wave, lane); break;
        }
        // fold limbs (exact in f64) and meet the other waves' K slices in LDS
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++)
            if (i < nt)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double v = (double)acc[i][mt][0][r] + 256.0 * (double)acc[i][mt][1][r] + 65536.0 * (double)acc[i][mt][2][r];
                        __hip_atomic_fetch_add(&accl[i][mt][r][lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
        __syncthreads();
        // epilogue: D[m][n] with n = lane & 15 (weight row of the tile), m = 4 * (lane >> 4) + reg (token in the tile)
        for (int e = threadIdx.x; e < nt * 2 * 4 * 64; e += SEQ_NT) {
            const int ln = e & 63, r = (e >> 6) & 3, mt = (e >> 8) & 1, i = e >> 9;
            const int t = mt * 16 + 4 * (ln >> 4) + r;
            const int id = tg + i, q = id / CB, cb = id % CB;
            const int row = Q * (16 * cb + (ln & 15)) + q;
            if (t < a.T && row < N) {
                const SeqVec rc = a.rec[v0 * SEQ_T + t];
                const double M = accl[i][mt][r][ln];
                const float v = (float)(rc.scale * (M + rc.cA + SEQ_CU * (double)a.rs[row])) + rc.So;
                const size_t o = (size_t)t * N + row;
                if (a.epi == 0) a.out[o] = v;
                else if (a.epi == 1) a.x[o] = (double)((float)a.x[o] + v);
                else {
                    const float gt = (float)(1.0 / (1.0 + exp(-(double)a.gate[(size_t)t * 5 * N + 5 * row + 4])));
                    a.x[o] = a.x[o] + (double)(v * gt);
                }
            }
        }
        __syncthreads();
        tg += nt;
    }
}

} // namespace rwkvk
