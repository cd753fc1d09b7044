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
//   k_seq_site    :412-465,:313-392 residual of the previous GEMM + LayerNorm + token-shift mix (the
//                                   shift runs along the chunk; token 0 takes the recurrent state) +
//                                   quantisation into the MFMA A-operand image
//   k_mm8_seq     :58-142,:267-311  [T x K] x [K x N] uint8 GEMM on MFMA, one launch per matrix group
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
    const double *xin;           // [T][D] residual stream before the pending residual update
    double *xout;                // [T][D] after it (a different buffer: neighbours read xin[t-1])
    int res_kind;                // 0 none; 1 x = f32(x) + add (att_out, rwkv.cu:548-553); 2 x += add * sigmoid(gate) (:574-577,:407)
    const float *add;            // GEMM output [T][ld_add], element [t][j]
    int ld_add;
    const float *gate;           // kind 2: ffn_r GEMM output, element [t][5 j + 4]
    int ld_gate;
    const double *lnw, *lnb;     // this site's LayerNorm rows
    const double *mix[3];        // token-shift mix per vector (nullptr: no shift, ln_out -> head)
    const float *r[3], *o[3];    // scale / offset of the matrices the vectors feed
    const double *state;         // previous LayerNorm output (state xy / dd, slot 0): token 0's shift input
    double *state_new;           // [D] LayerNorm output of the last token (copied over the state afterwards)
    unsigned *img[3];            // A-operand images
    SeqVec *rec;                 // [NV][SEQ_T]
    int D, T;
};

constexpr int SEQ_NQ = 3;        // quads per thread: D <= 5120 -> 1280 quads over 512 threads

template <int NV>
__global__ __launch_bounds__(NT) void k_seq_site(SeqSiteArgs a)
{
    __shared__ double red[RED_BYTES / 8];
    __shared__ unsigned lds_sums[12];
    const int D = a.D, t = blockIdx.x, nqd = D >> 2;
    const bool shift = a.mix[0] != nullptr;
    auto load_x = [&](int tt, int j) -> double {
        const double x0 = a.xin[(size_t)tt * D + j];
        if (a.res_kind == 1) return (double)((float)x0 + a.add[(size_t)tt * a.ld_add + j]);
        if (a.res_kind == 2) {
            const float g = (float)(1.0 / (1.0 + exp(-(double)a.gate[(size_t)tt * a.ld_gate + 5 * j + 4])));   // rwkv.cu:212
            return x0 + (double)(a.add[(size_t)tt * a.ld_add + j] * g);                                         // :407
        }
        return x0;
    };
    double xt[SEQ_NQ][4], xp[SEQ_NQ][4];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < SEQ_NQ; i++) {
        const int qd = threadIdx.x + i * NT;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xt[i][e] = 0.0; xp[i][e] = 0.0;
            if (qd < nqd) {
                const int j = qd * 4 + e;
                xt[i][e] = load_x(t, j);
                a.xout[(size_t)t * D + j] = xt[i][e];
                s[0] += xt[i][e]; s[1] += xt[i][e] * xt[i][e];
                if (shift) {
                    if (t > 0) { xp[i][e] = load_x(t - 1, j); s[2] += xp[i][e]; s[3] += xp[i][e] * xp[i][e]; }
                    else xp[i][e] = a.state[j];
                }
            }
        }
    }
    block_sum<4>(s, red + RED_STATS);
    const double mean = s[0] / (double)D, rstd = 1.0 / sqrt((s[1] - s[0] * mean) / (double)(D - 1));
    const double meanp = s[2] / (double)D, rstdp = (shift && t > 0) ? 1.0 / sqrt((s[3] - s[2] * meanp) / (double)(D - 1)) : 1.0;

    float xr[NV][SEQ_NQ][4];
    double So[NV];
    float amax[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { So[m] = 0.0; amax[m] = 0.f; }
#pragma unroll
    for (int i = 0; i < SEQ_NQ; i++) {
        const int qd = threadIdx.x + i * NT;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = qd * 4 + e;
            const bool real = qd < nqd;
            const int jc = real ? j : 0;
            const double xx = a.lnw[jc] * ((xt[i][e] - mean) * rstd) + a.lnb[jc];
            double xprev = xp[i][e];
            if (shift && t > 0) xprev = a.lnw[jc] * ((xp[i][e] - meanp) * rstdp) + a.lnb[jc];
            if (real && a.state_new && t == a.T - 1) a.state_new[j] = xx;          // mixatt / mixffn state write (:344,:385)
#pragma unroll
            for (int m = 0; m < NV; m++) {
                float f = (float)xx;
                if (shift) { const double mk = a.mix[m][jc]; f = (float)(xx * mk + xprev * (1.0 - mk)); }   // :339-343,:377-384
                const float v = real ? f * a.r[m][jc] : 0.f;
                xr[m][i][e] = v;
                if (real) { So[m] += (double)(f * a.o[m][jc]); amax[m] = fmaxf(amax[m], fabsf(v)); }
            }
        }
    }
    block_sum<NV>(So, red + RED_OFFS);
    block_max<NV>(amax, red + RED_MAX);
    unsigned ls[NV][3];
#pragma unroll
    for (int m = 0; m < NV; m++) {
        ls[m][0] = ls[m][1] = ls[m][2] = 0u;
        const float inv_s = inv_scale(amax[m]);
#pragma unroll
        for (int i = 0; i < SEQ_NQ; i++) {
            const int qd = threadIdx.x + i * NT;
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
                const int kk = qd * 4 + e;
                float f;
                if (KIND == 0) f = a.src[(size_t)t * K + kk];
                else {
                    const float v = a.src[(size_t)t * (K / 4 * 5) + (kk >> 2) * 5 + (kk & 3)];
                    f = v * (float)(v > 0.f);
                    f = f * f;
                }
                xr[i][e] = f * a.r[kk];
                So[0] += (double)(f * a.o[kk]);
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
};
// one thread per channel, sequential over the chunk (rwkv.cu:242-255 with the GPT-mode state slot 0)
__global__ void k_seq_wkv(SeqWkvArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.D) return;
    double aa = a.saa[i], bb = a.sbb[i];
    const double uw = a.uw[i], ew = a.ew[i];
    for (int t = 0; t < a.T; t++) {
        const float *p = a.kvr + (size_t)t * 3 * a.D + 3 * i;
        const float k = p[0], v = p[1], r = p[2];
        const double vv = (double)v;
        const double e1 = exp(uw + (double)k);
        double y = (aa + e1 * vv) / (bb + e1);
        y = (1.0 / (1.0 + (double)expf(-r))) * y;
        const double ek = exp((double)k);
        aa = (aa + ek * vv) * ew;
        bb = (bb + ek) * ew;
        a.y[(size_t)t * a.D + i] = (float)y;
    }
    a.saa[i] = aa; a.sbb[i] = bb;
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
    float *out;                  // [T][N]
    int T;
};
constexpr int SEQ_TB = 5;        // weight tiles (16 rows each) per workgroup pass

// One workgroup owns a contiguous range of (class-major) 16-row tiles; the 8 waves split K; the
// integer partial sums meet in LDS (exact: order does not matter), then the workgroup applies scale,
// offsets and corrections and writes f32.  Weight bytes are read once, with 16 B per lane
// non-temporal loads: lane (g, c) reads bytes [64 kb + 16 g, +16) of row c of the tile, which is the
// register image of the B operand.
__global__ __launch_bounds__(NT) void k_mm8_seq(SeqGemmArgs a)
{
    __shared__ double accl[SEQ_TB][2][4][64];   // [tile][token tile][reg][lane], 20 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nch = (N + Q - 1) / Q;                    // channels per class
    const int CB = (nch + 15) >> 4;                     // 16-channel blocks per class
    const int ntiles = Q * CB;
    const int tb0 = (int)(((long long)blockIdx.x * ntiles) / gridDim.x), tb1 = (int)(((long long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    const int kb0 = (wave * KB) / NW, kb1 = ((wave + 1) * KB) / NW;
    const int g = lane >> 4, c = lane & 15;

    for (int tg = tb0; tg < tb1; tg += SEQ_TB) {
        const int nt = tb1 - tg < SEQ_TB ? tb1 - tg : SEQ_TB;
        for (int e = threadIdx.x; e < SEQ_TB * 2 * 4 * 64; e += NT) (&accl[0][0][0][0])[e] = 0.0;
        __syncthreads();

        // per tile: weight row of this lane's column, and the activation vector
        const uint8_t *wrow[SEQ_TB];
        int vec[SEQ_TB];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++) {
            const int id = tg + (i < nt ? i : 0);
            const int q = id / CB, cb = id % CB;
            int row = Q * (16 * cb + c) + q;
            row = row < N ? row : N - 1;
            wrow[i] = a.w + (size_t)row * K + 16 * g;
            vec[i] = a.vec_of_q[q];
        }
        i32x4 acc[SEQ_TB][2][3];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int b = 0; b < 3; b++) acc[i][mt][b] = i32x4{0, 0, 0, 0};

        for (int kb = kb0; kb < kb1; kb++) {
            u32x4 bw[SEQ_TB];
#pragma unroll
            for (int i = 0; i < SEQ_TB; i++) bw[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow[i] + (size_t)kb * 64));
            i32x4 af[2][3];
            int cur = -1;
#pragma unroll
            for (int i = 0; i < SEQ_TB; i++) {
                if (i < nt) {
                    if (vec[i] != cur) {   // wave-uniform: tiles of one class are contiguous
                        cur = vec[i];
#pragma unroll
                        for (int mt = 0; mt < 2; mt++)
#pragma unroll
                            for (int b = 0; b < 3; b++) {
                                const u32x4 t4 = a.img[cur][a_unit(kb, mt, b, lane)];
                                af[mt][b] = i32x4{(int)t4[0], (int)t4[1], (int)t4[2], (int)t4[3]};
                            }
                    }
                    const u32x4 x4 = bw[i] ^ 0x80808080u;
                    const i32x4 bf = i32x4{(int)x4[0], (int)x4[1], (int)x4[2], (int)x4[3]};
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int b = 0; b < 3; b++)
                            acc[i][mt][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[mt][b], bf, acc[i][mt][b], 0, 0, 0);
                }
            }
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
        for (int e = threadIdx.x; e < nt * 2 * 4 * 64; e += NT) {
            const int ln = e & 63, r = (e >> 6) & 3, mt = (e >> 8) & 1, i = e >> 9;
            const int t = mt * 16 + 4 * (ln >> 4) + r;
            const int id = tg + i, q = id / CB, cb = id % CB;
            const int row = Q * (16 * cb + (ln & 15)) + q;
            if (t < a.T && row < N) {
                const SeqVec rc = a.rec[a.vec_of_q[q] * SEQ_T + t];
                const double M = accl[i][mt][r][ln];
                a.out[(size_t)t * N + row] = (float)(rc.scale * (M + rc.cA + SEQ_CU * (double)a.rs[row])) + rc.So;
            }
        }
        __syncthreads();
    }
}

} // namespace rwkvk
