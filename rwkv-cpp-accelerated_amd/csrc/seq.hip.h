// seq.hip.h -- gfx950 device code of the CHUNKED (prompt prefill / batched) forward: up to 32 rows per pass, every
// uint8 weight byte read ONCE per chunk instead of once per token.
//
// Replaces, for MODE GPT with several tokens, the reference's per-op loops over `tokenlength` (rwkv.cu:493-593 with
// tokenlength > 1; mm8 kernels :58-142,:267-311 re-stream the weights per token), i.e. what RWKV::loadContext drives
// (rwkv.h:395-413), and for MODE PARRALEL one step of up to 32 independent sequences.  Same arithmetic contract as the
// single-token kernels (kernels.hip.h): activations are quantised per (vector, row) to 23-bit fixed point, split into
// three byte limbs, and contracted against the uint8 weights in exact integer arithmetic -- here on the matrix cores:
// v_mfma_i32_16x16x64_i8 (a real int8 contraction: M = 16 rows of the chunk, N = 16 weight rows, K = 64).  Both operands
// are stored signed (byte - 128); the offsets are removed with the weight row sums and the per-row limb sums:
//     sum_k u_k l_k = MFMA(u - 128, l - 128) + 128 * rowsum(u) + 128 * sum_k (l_k - 128).
//
// Round 2 layout (round 1 ran the element-wise work with one workgroup per chunk row -- 32 workgroups on a 256-CU part --
// and let every GEMM workgroup re-read the whole activation image):
//   * weights have a second resident copy in the MFMA B-OPERAND IMAGE [16-row tile][k-block of 64][lane][16 B], signed,
//     so a wave's weight load is one contiguous 1 KiB and needs no cross-lane transposition (7 GB more at 7B of 288 GB);
//   * k_seq_gemm_ks (K/V/R, ffn k/r, head: many rows per workgroup): 8 waves take the k-blocks round-robin over the
//     workgroup's <= 5 row tiles, operands two k-blocks ahead, exact f64 meeting in LDS, epilogue in the kernel;
//   * k_seq_gemm_tw (att_out, ffn_v: 4096 rows only, where every workgroup re-reading the whole activation image moved
//     6x the weight bytes through L2): 8 K-SLICES, slice j on XCD j (blockIdx % 8), a workgroup = 128 rows x one slice,
//     one row tile per wave, the slice's activation image staged once in LDS for all 8 waves; exact f64 partial sums
//     [8][T][N] that the following element-wise kernel adds up;
//   * element-wise work runs on (row, octant) workgroups -- 256 of them for a full chunk: k_seq_resid (partial sums ->
//     residual update + LayerNorm partial statistics), k_seq_site (LayerNorm + token shift + quantisation with the
//     decode path's upper bound for the fixed-point scale, so ONE statistics round suffices), k_seq_stage (relu^2 / gated
//     wkv -> quantisation), k_seq_wkv (the recurrence).
//
//   k_seq_embed   rwkv.cu:513-524   embedding rows + ln0 for the chunk
//   k_seq_resid   :548-553,:574-577,:407   residual updates behind att_out / ffn_v; LayerNorm statistics (:40-57)
//   k_seq_site    :412-465,:313-392 LayerNorm + token-shift mix (the shift runs along the chunk; row 0 takes the
//                                   recurrent state) + quantisation into the MFMA A-operand image
//   k_seq_gemm_*  :58-142,:267-311  [T x K] x [K x N] uint8 GEMM on MFMA
//   k_seq_wkv     :221-259          WKV recurrence, sequential over the chunk per channel
//   k_seq_stage   :144-219          relu^2 / cast + scale + quantisation for att_out and ffn_v inputs
#pragma once
#include "kernels.hip.h"

namespace rwkvk {

constexpr int SEQ_T = 32;                 // rows per pass: two 16-row MFMA tiles
constexpr int SEQ_O = 8;                  // octants of a row (element-wise workgroups per row) = K-slices = XCDs
constexpr double SEQ_CU = 4227200.0;      // 128 * (1 + 256 + 65536) - 2^22: weight-row-sum coefficient
typedef int i32x4 __attribute__((ext_vector_type(4)));

// A-operand image of one activation vector for a chunk: 16-byte units indexed
// [k-block of 64][row tile of 16][limb][lane], lane = 16 * ((k % 64) / 16) + row % 16 -- exactly the
// register image of v_mfma_i32_16x16x64_i8's A operand, so a wave loads a fragment with one
// coalesced 1 KiB global_load_dwordx4.
__device__ __forceinline__ size_t a_unit(int kb, int mt, int limb, int lane) { return (((size_t)kb * 2 + mt) * 3 + limb) * 64 + lane; }
__host__ __device__ inline size_t a_image_bytes(size_t K) { return (K / 64) * 2 * 3 * 64 * 16; }

// quantisation record of one (vector, row): scale from the row's amax, the limb / offset sums as per-octant partials
struct SeqPart { double cA; float So; float pad; };       // cA: 128 * sum_b 256^b * sum_k (limb_b[k] - 128) over the octant; So: sum_k f[k] * o[k]
struct SeqStat { double sx, sxx; float amax, pmax; };     // per (row, octant): sum x, sum x^2, max|x|, max|shift source| (state rows only)
struct SeqRec { double scale, cA; float So; };
__device__ __forceinline__ SeqRec seq_rec(const SeqPart *part, const float *amaxv, int m, int t)
{
    SeqRec r;
    r.scale = scale_of(amaxv[m * SEQ_T + t]);
    double ca = 0.0;
    float so = 0.f;
    const SeqPart *p = part + ((size_t)m * SEQ_T + t) * SEQ_O;
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) { ca += p[o].cA; so += p[o].So; }
    r.cA = ca; r.So = so;
    return r;
}
// octant o of K elements, in units of 64 (a k-block never straddles two octants): [k0, k1)
__device__ __forceinline__ void octant_range(int K, int o, int &k0, int &k1)
{
    const int KB = K >> 6;
    k0 = (int)(((long long)o * KB) / SEQ_O) << 6;
    k1 = (int)(((long long)(o + 1) * KB) / SEQ_O) << 6;
}

// quantise one quad (4 consecutive k) of row t and store its three signed limb dwords; adds the sums of the unsigned
// limb bytes to s[0..2]
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

constexpr int SEQ_ENT = 256;              // threads of an element-wise (row, octant) workgroup
constexpr int SEQ_ENW = SEQ_ENT / 64;
// workgroup sums / maxima for SEQ_ENT threads; `red` = SEQ_ENW * K doubles of LDS, one barrier pair per call
template <int K>
__device__ __forceinline__ void eblock_sum(double (&v)[K], double *red)
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
        for (int i = 0; i < SEQ_ENW; i++) t += red[i * K + k];
        v[k] = t;
    }
}
template <int K>
__device__ __forceinline__ void eblock_max(float (&v)[K], double *redd)
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
        for (int i = 0; i < SEQ_ENW; i++) t = fmaxf(t, red[i * K + k]);
        v[k] = t;
    }
}
// limb sums + offset sum of one (vector, row, octant) -> its SeqPart; nk = elements of the octant
__device__ __forceinline__ void seq_finish(unsigned (&ls)[3], double So, int nk, SeqPart *dst, double *red)
{
    double s[4] = {(double)ls[0], (double)ls[1], (double)ls[2], So};   // the limb sums of a thread are < 2^32: exact in f64
    eblock_sum<4>(s, red);
    if (threadIdx.x == 0) {
        double ca = 0.0, f = 128.0;
#pragma unroll
        for (int b = 0; b < 3; b++) { ca += f * (s[b] - 128.0 * (double)nk); f *= 256.0; }
        SeqPart p;
        p.cA = ca; p.So = (float)s[3]; p.pad = 0.f;
        *dst = p;
    }
}

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
// (row, octant) workgroups: fold the K-slice partial sums of att_out / ffn_v into the residual stream and emit the
// LayerNorm partial statistics of the new x.  MODE 0: statistics only (first layer of a chunk); 1: att_out
// (x = f32(x) + v, rwkv.cu:548-553); 2: ffn_v (x += v * sigmoid(r), :574-577,:407,:212).
struct SeqResidArgs {
    double *x;                   // [T][D]
    const double *part;          // [SEQ_O][T][D] exact integer partial sums of the GEMM (K-slice major)
    const unsigned *rs;          // [D] weight row sums
    const SeqPart *qpart;        // records of the GEMM's input vector (vector 0): [T][SEQ_O]
    const float *amaxv;          // [T]
    const float *gate;           // MODE 2: ffn k/r GEMM output [T][5D], r of channel j at [t][5 j + 4]
    const double *state;         // shift source of the NEXT site for rows that take it from the recurrent state:
    size_t slot_stride;          //   GPT: row 0 <- state (slot 0); PARRALEL: row t <- state + (slot0 + t) * slot_stride; nullptr: none
    int slot0, par;
    SeqStat *stat;               // [T][SEQ_O]
    int D, T;
};
template <int MODE>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_resid(SeqResidArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    const int D = a.D, t = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;
    int c0, c1;
    octant_range(D, o, c0, c1);
    SeqRec rc{};
    if (MODE != 0) rc = seq_rec(a.qpart, a.amaxv, 0, t);
    const double *st = nullptr;
    if (a.state) st = a.par ? a.state + (size_t)(a.slot0 + t) * a.slot_stride : (t == 0 ? a.state : nullptr);
    double s[2] = {0.0, 0.0};
    float mx[2] = {0.f, 0.f};
    for (int j = c0 + threadIdx.x; j < c1; j += SEQ_ENT) {
        const size_t e = (size_t)t * D + j;
        double x = a.x[e];
        if (MODE != 0) {
            double M = 0.0;
#pragma unroll
            for (int q = 0; q < SEQ_O; q++) M += a.part[((size_t)q * SEQ_T + t) * D + j];
            const float v = (float)(rc.scale * (M + rc.cA + SEQ_CU * (double)a.rs[j])) + rc.So;
            if (MODE == 1) x = (double)((float)x + v);
            else {
                const float gt = (float)(1.0 / (1.0 + exp(-(double)a.gate[(size_t)t * 5 * D + 5 * (size_t)j + 4])));
                x = x + (double)(v * gt);
            }
            a.x[e] = x;
        }
        s[0] += x; s[1] += x * x;
        mx[0] = fmaxf(mx[0], (float)fabs(x) * 1.0000002f);
        if (st) mx[1] = fmaxf(mx[1], (float)fabs(st[j]) * 1.0000002f);
    }
    eblock_sum<2>(s, red);
    eblock_max<2>(mx, red);
    if (threadIdx.x == 0) {
        SeqStat r;
        r.sx = s[0]; r.sxx = s[1]; r.amax = mx[0]; r.pmax = mx[1];
        a.stat[t * SEQ_O + o] = r;
    }
}

// ------------------------------------------------------------------------------------------
struct SeqSiteArgs {
    const double *x;             // [T][D] residual stream
    const SeqStat *stat;         // [T][SEQ_O]
    const double *lnw, *lnb;     // this site's LayerNorm rows
    const double *mix[3];        // token-shift mix per vector (nullptr: no shift, ln_out -> head)
    const float *r[3], *o[3];    // scale / offset of the matrices the vectors feed
    const float *mxs;            // static bounds of the site: per vector {max|r mix lnw|, max|r mix lnb|, max|r (1 - mix)|, 0}, then {max|lnw|, max|lnb|}
    const double *state;         // previous LayerNorm output (state xy / dd of this layer, slot 0): row 0's shift input
    double *state_new;           // GPT: [D] LayerNorm output of the last row (copied over the state afterwards)
    int par;                     // PARRALEL mode (rwkv.cu:236-240): row t is an independent sequence with state slot slot0 + t --
    double *state_par;           //   shift input and state write both go to that slot of this array (same base as `state`)
    size_t slot_stride;          //   L * D
    int slot0;
    unsigned *img[3];            // A-operand images
    SeqPart *part;               // [NV][T][SEQ_O]
    float *amaxv;                // [NV][T]
    int D, T;
};
// mean, rstd, max|x| of row t from its octant partials (every thread computes them: 8 tiny loads)
__device__ __forceinline__ void seq_row_stats(const SeqStat *stat, int t, int D, double &mean, double &rstd, float &xmax, float &pmax)
{
    double sx = 0.0, sxx = 0.0;
    xmax = 0.f; pmax = 0.f;
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) {
        const SeqStat r = stat[t * SEQ_O + o];
        sx += r.sx; sxx += r.sxx; xmax = fmaxf(xmax, r.amax); pmax = fmaxf(pmax, r.pmax);
    }
    mean = sx / (double)D;
    rstd = 1.0 / sqrt((sxx - sx * mean) / (double)(D - 1));    // reference: (D-1), no epsilon (rwkv.cu:43-44,53)
}
// (row, octant) workgroups.  The fixed-point scale of vector m comes from an UPPER BOUND of max|v_m| that needs only the
// row statistics (as the decode kernels' site_reduce): |v| <= max|r mix lnw| (max|x| + |mean|) rstd + max|r mix lnb|
// + max|r (1 - mix)| max|shift source|, so the site needs one statistics round, not two.
template <int NV>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_site(SeqSiteArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    const int D = a.D, t = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;
    int c0, c1;
    octant_range(D, o, c0, c1);
    const bool shift = a.mix[0] != nullptr;
    const bool lnprev = shift && t > 0 && !a.par;   // the shift input is a LayerNorm output: of the previous row (GPT) or already stored (state)
    const double *xprow = !shift ? a.x : a.par ? a.state_par + (size_t)(a.slot0 + t) * a.slot_stride : (t > 0 ? a.x + (size_t)(t - 1) * D : a.state);
    double mean, rstd, meanp = 0.0, rstdp = 1.0;
    float xmax, pmax, xmaxp = 0.f, dummy;
    seq_row_stats(a.stat, t, D, mean, rstd, xmax, pmax);
    if (lnprev) seq_row_stats(a.stat, t - 1, D, meanp, rstdp, xmaxp, dummy);
    const float mlw = a.mxs[4 * NV], mlb = a.mxs[4 * NV + 1];
    const float xhmax = (float)(((double)xmax + fabs(mean)) * rstd) * 1.0001f;
    const float prevmax = !shift ? 0.f : lnprev ? (mlw * (float)(((double)xmaxp + fabs(meanp)) * rstdp) * 1.0001f + mlb) : pmax;
    float amax[NV], inv_s[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) {
        amax[m] = (a.mxs[4 * m] * xhmax + a.mxs[4 * m + 1] + a.mxs[4 * m + 2] * prevmax) * 1.0001f;
        inv_s[m] = inv_scale(amax[m]);
    }
    unsigned ls[NV][3];
    double So[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { ls[m][0] = ls[m][1] = ls[m][2] = 0u; So[m] = 0.0; }
    const int q0 = c0 >> 2, q1 = c1 >> 2;
    for (int qd = q0 + threadIdx.x; qd < q1; qd += SEQ_ENT) {
        double xt[4], xp[4], lw[4], lb[4];
        load_quad_f64(a.x + (size_t)t * D, qd, xt);
        load_quad_f64(xprow, qd, xp);
        load_quad_f64(a.lnw, qd, lw);
        load_quad_f64(a.lnb, qd, lb);
        double xx[4], xprev[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xx[e] = lw[e] * ((xt[e] - mean) * rstd) + lb[e];
            xprev[e] = lnprev ? lw[e] * ((xp[e] - meanp) * rstdp) + lb[e] : xp[e];
            const int j = qd * 4 + e;
            if (a.par) a.state_par[(size_t)(a.slot0 + t) * a.slot_stride + j] = xx[e];   // own slot: read above by this thread only
            else if (a.state_new && t == a.T - 1) a.state_new[j] = xx[e];                // mixatt / mixffn state write (:344,:385)
        }
#pragma unroll
        for (int m = 0; m < NV; m++) {
            const f32x4 rr = reinterpret_cast<const f32x4 *>(a.r[m])[qd], oo = reinterpret_cast<const f32x4 *>(a.o[m])[qd];
            double mk[4] = {1.0, 1.0, 1.0, 1.0};
            if (shift) load_quad_f64(a.mix[m], qd, mk);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float f = (float)xx[e];
                if (shift) f = (float)(xx[e] * mk[e] + xprev[e] * (1.0 - mk[e]));   // :339-343,:377-384
                v[e] = f * rr[e];
                So[m] += (double)(f * oo[e]);
            }
            seq_store_quad(a.img[m], qd, t, v, inv_s[m], ls[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < NV; m++) {
        seq_finish(ls[m], So[m], c1 - c0, a.part + ((size_t)m * SEQ_T + t) * SEQ_O + o, red);
        if (o == 0 && threadIdx.x == 0) a.amaxv[m * SEQ_T + t] = amax[m];
    }
}

// ------------------------------------------------------------------------------------------
struct SeqStageArgs {
    const float *src;            // kind 0: gated wkv y [T][D]; kind 1: ffn k/r GEMM output [T][5D], element [t][5 i + q]
    const float *r, *o;          // scale / offset over K
    const float *pmax;           // [n_pmax][SEQ_T] per-producer-workgroup partial max of |f * r| per row
    int n_pmax;
    unsigned *img;
    SeqPart *part;               // [T][SEQ_O]
    float *amaxv;                // [T]
    int K, T;
};
// (row, octant) workgroups.  KIND 0: f = src (att_out input).  KIND 1: f = relu(k)^2 with k = ffn_k GEMM output
// (rwkv.cu:189-190), K = 4D.  The exact max|f * r| of the row comes from the producers' partials.
template <int KIND>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_stage(SeqStageArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    const int K = a.K, t = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;
    int c0, c1;
    octant_range(K, o, c0, c1);
    float am[1] = {0.f};
    for (int w = threadIdx.x; w < a.n_pmax; w += SEQ_ENT) am[0] = fmaxf(am[0], a.pmax[(size_t)w * SEQ_T + t]);
    eblock_max<1>(am, red);
    const float inv_s = inv_scale(am[0]);
    unsigned ls[3] = {0u, 0u, 0u};
    double So = 0.0;
    const int q0 = c0 >> 2, q1 = c1 >> 2;
    for (int qd = q0 + threadIdx.x; qd < q1; qd += SEQ_ENT) {
        f32x4 sv;
        if (KIND == 0) sv = reinterpret_cast<const f32x4 *>(a.src + (size_t)t * K)[qd];
        else __builtin_memcpy(&sv, a.src + (size_t)t * (K / 4 * 5) + (size_t)qd * 5, 16);   // k0..k3 of channel qd: 4-byte aligned only
        const f32x4 rv = reinterpret_cast<const f32x4 *>(a.r)[qd], ov = reinterpret_cast<const f32x4 *>(a.o)[qd];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float f = sv[e];
            if (KIND == 1) { f = f * (float)(f > 0.f); f = f * f; }     // relu(k)^2, rwkv.cu:189-190
            v[e] = f * rv[e];
            So += (double)(f * ov[e]);
        }
        seq_store_quad(a.img, qd, t, v, inv_s, ls);
    }
    seq_finish(ls, So, c1 - c0, a.part + (size_t)t * SEQ_O + o, red);
    if (o == 0 && threadIdx.x == 0) a.amaxv[t] = am[0];
}

// ------------------------------------------------------------------------------------------
struct SeqWkvArgs {
    const float *kvr;            // K/V/R GEMM output [T][3D], element [t][3 i + m]
    const double *uw, *ew;       // bonus + decay, exp(decay) of this layer
    double *saa, *sbb;           // state of this layer, slot 0
    const float *r_att;          // att_out scale [D]: the partial maxima are those of y * r_att (att_out's staged vector)
    float *y;                    // [T][D] gated wkv, cast to f32 as the att_out GEMV does (rwkv.cu:290)
    float *pmax;                 // [gridDim.x][SEQ_T]
    int D, T;
    int par;                     // PARRALEL mode: row t uses state slot slot0 + t (no recurrence along the rows)
    size_t slot_stride;
    int slot0;
};
constexpr int WKV_CH = 16;       // channels per workgroup (512 threads = 16 channels x 32 rows)
// rwkv.cu:242-255 with the GPT-mode state slot 0.  The exponentials do not depend on the state, so
// one thread per (row, channel) evaluates them; then one thread per channel runs the recurrence
// along the chunk (a division and a few fma per step).
__global__ __launch_bounds__(SEQ_T * WKV_CH) void k_seq_wkv(SeqWkvArgs a)
{
    __shared__ double e1s[SEQ_T][WKV_CH], eks[SEQ_T][WKV_CH], vs[SEQ_T][WKV_CH], sgs[SEQ_T][WKV_CH];
    __shared__ float ya[SEQ_T][WKV_CH];
    const int ch = threadIdx.x & (WKV_CH - 1), t = threadIdx.x / WKV_CH;
    const int i = blockIdx.x * WKV_CH + ch;
    const bool live = i < a.D && t < a.T;
    ya[t][ch] = 0.f;
    if (live) {
        const float *p = a.kvr + (size_t)t * 3 * a.D + 3 * i;
        const float k = p[0], v = p[1], r = p[2];
        e1s[t][ch] = exp(a.uw[i] + (double)k);
        eks[t][ch] = exp((double)k);
        vs[t][ch] = (double)v;
        sgs[t][ch] = 1.0 / (1.0 + (double)expf(-r));       // rwkv.cu:250: exp of a float argument
    }
    __syncthreads();
    if (a.par) {
        if (live) {
            const size_t so = (size_t)(a.slot0 + t) * a.slot_stride + i;
            const double aa = a.saa[so], bb = a.sbb[so], ew = a.ew[i];
            const double e1 = e1s[t][ch], ek = eks[t][ch], vv = vs[t][ch];
            const float yf = (float)(sgs[t][ch] * ((aa + e1 * vv) / (bb + e1)));
            a.y[(size_t)t * a.D + i] = yf;
            ya[t][ch] = fabsf(yf * a.r_att[i]);
            a.saa[so] = (aa + ek * vv) * ew;
            a.sbb[so] = (bb + ek) * ew;
        }
    } else if (threadIdx.x < WKV_CH && i < a.D) {
        double aa = a.saa[i], bb = a.sbb[i];
        const double ew = a.ew[i];
        const float ra = a.r_att[i];
        for (int tt = 0; tt < a.T; tt++) {
            const double e1 = e1s[tt][ch], ek = eks[tt][ch], vv = vs[tt][ch];
            const double y = sgs[tt][ch] * ((aa + e1 * vv) / (bb + e1));
            aa = (aa + ek * vv) * ew;
            bb = (bb + ek) * ew;
            const float yf = (float)y;
            a.y[(size_t)tt * a.D + i] = yf;
            ya[tt][ch] = fabsf(yf * ra);
        }
        a.saa[i] = aa; a.sbb[i] = bb;
    }
    __syncthreads();
    if (threadIdx.x < SEQ_T) {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < WKV_CH; c++) m = fmaxf(m, ya[threadIdx.x][c]);
        a.pmax[(size_t)blockIdx.x * SEQ_T + threadIdx.x] = m;
    }
}

// ------------------------------------------------------------------------------------------
// Load-time: MFMA B-operand image of a re-tiled matrix w_t[N][K] (row-per-output).  Tiles are enumerated class-major:
// tile id -> class q = id / CB, 16-channel block cb = id % CB, tile row c -> matrix row Q * (16 cb + c) + q (Q row classes
// interleaved in w_t: K/V/R 3, ffn k,k,k,k,r 5, else 1), so that a pass of consecutive tiles shares one activation vector.
// bimg[((id * KB + kb) * 64 + lane) * 16 + b] = w_t[row(id, lane & 15)][64 kb + 16 (lane >> 4) + b] - 128 (rows past N: 0).
__global__ void k_bimage(const uint8_t *__restrict__ w_t, uint8_t *__restrict__ bimg, int N, int K, int Q, int CB)
{
    const int KB = K >> 6;
    const size_t unit = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte unit per thread
    const size_t total = (size_t)Q * CB * KB * 64;
    if (unit >= total) return;
    const int lane = (int)(unit & 63);
    const size_t tk = unit >> 6;
    const int kb = (int)(tk % KB), id = (int)(tk / KB);
    const int q = id / CB, cb = id % CB;
    const int row = Q * (16 * cb + (lane & 15)) + q;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (row < N) {
        v = *reinterpret_cast<const u32x4 *>(w_t + (size_t)row * K + (size_t)kb * 64 + 16 * (lane >> 4));
#pragma unroll
        for (int d = 0; d < 4; d++) v[d] ^= 0x80808080u;
    }
    reinterpret_cast<u32x4 *>(bimg)[unit] = v;
}

// ------------------------------------------------------------------------------------------
struct SeqGemmArgs {
    const u32x4 *bimg;           // B-operand image of the weights (k_bimage)
    const unsigned *rs;          // [N] row sums of the unsigned weights
    int N, K;
    int Q;                       // row classes interleaved in the matrix
    int vec_of_q[5];             // activation vector each class multiplies
    const u32x4 *img[3];         // A-operand images of the vectors
    const SeqPart *part;         // quantisation records [NV][T][SEQ_O]
    const float *amaxv;          // [NV][T]
    float *out;                  // epi 0 / 3: [T][N] f32
    int epi;                     // 0 store; 3 store + per-workgroup partial max of relu(k)^2 * r_fv per row (ffn k/r GEMM, classes 0..3)
    const float *r_next;         // epi 3: ffn_v scale [4 nch], hidden unit 4 i + q
    float *pmax;                 // epi 3: [gridDim.x][SEQ_T]
    int T;
    const double *cp_src;        // piggy-back copy (stream-ordered behind the site kernel that produced it): the chunk's
    double *cp_dst;              // last LayerNorm output -> recurrent state; cp_n == 0: none
    int cp_n;
};
constexpr int SEQ_TB = 5;        // weight tiles (16 rows each) per workgroup pass
constexpr int SEQ_NT = 512;      // GEMM workgroup: 8 waves, two per SIMD
constexpr int SEQ_NW = SEQ_NT / 64;

// operands of one k-block: B fragments of NTL tiles and the A fragments of the pass's vector
template <int NTL> struct SeqFrag { u32x4 bw[NTL]; u32x4 af[2][3]; };
template <int NTL>
__device__ __forceinline__ void seq_frag_load(SeqFrag<NTL> &f, const u32x4 *const (&wt)[SEQ_TB], const u32x4 *img, int kb, int lane)
{
#pragma unroll
    for (int i = 0; i < NTL; i++) f.bw[i] = __builtin_nontemporal_load(wt[i] + (size_t)kb * 64);
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int b = 0; b < 3; b++) f.af[mt][b] = img[a_unit(kb, mt, b, lane)];
}
template <int NTL>
__device__ __forceinline__ void seq_frag_mfma(const SeqFrag<NTL> &f, i32x4 (&acc)[SEQ_TB][2][3])
{
#pragma unroll
    for (int i = 0; i < NTL; i++) {
        const i32x4 bf = i32x4{(int)f.bw[i][0], (int)f.bw[i][1], (int)f.bw[i][2], (int)f.bw[i][3]};
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const i32x4 af = i32x4{(int)f.af[mt][b][0], (int)f.af[mt][b][1], (int)f.af[mt][b][2], (int)f.af[mt][b][3]};
                acc[i][mt][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[i][mt][b], 0, 0, 0);
            }
    }
}
// K loop of one pass over NTL tiles: the waves take k-blocks round-robin (wave w: kb = w, w + 8, ...), so at any moment the
// workgroup reads 8 adjacent 1 KiB pieces of each tile's image; operands of the next k-block(s) are requested before
// the MFMAs of the current one issue.
template <int NTL, int DEPTH>
__device__ __forceinline__ void seq_pass(i32x4 (&acc)[SEQ_TB][2][3], const u32x4 *const (&wt)[SEQ_TB], const u32x4 *img, int KB, int wave, int lane)
{
    const int n = (KB - wave + SEQ_NW - 1) / SEQ_NW;      // k-blocks of this wave
    if (n <= 0) return;
    auto kbi = [&](int it) { return wave + SEQ_NW * (it < n - 1 ? it : n - 1); };   // past the end: re-read the last (branch-free loop)
    SeqFrag<NTL> f[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++) seq_frag_load<NTL>(f[j], wt, img, kbi(j), lane);
    int it = 0;
    for (; it + DEPTH <= n; it += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            seq_frag_load<NTL>(f[(j + DEPTH - 1) % DEPTH], wt, img, kbi(it + j + DEPTH - 1), lane);
            seq_frag_mfma<NTL>(f[j], acc);
        }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++)
        if (it + j < n) seq_frag_mfma<NTL>(f[j], acc);
}

// "K over the waves": one workgroup owns a contiguous range of (class-major) 16-row tiles and works through it in passes
// of up to SEQ_TB tiles that share an activation vector; its 8 waves split K; the integer partial sums meet in LDS
// (exact: order does not matter), then the workgroup applies scale, offsets and corrections.
// TAG only names the launch for the profiler: 0 K/V/R, 1 ffn k/r, 2 head
template <int TAG>
__global__ __launch_bounds__(SEQ_NT) void k_seq_gemm_ks(SeqGemmArgs a)
{
    __shared__ double accl[SEQ_TB][2][4][64];   // [tile][row tile][reg][lane], 20 KiB
    __shared__ SeqRec recl[SEQ_T];
    __shared__ float pml[SEQ_T];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nch = (N + Q - 1) / Q;                    // channels per class
    const int CB = (nch + 15) >> 4;                     // 16-channel blocks per class
    const int ntiles = Q * CB;
    const int tb0 = (int)(((long long)blockIdx.x * ntiles) / gridDim.x), tb1 = (int)(((long long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    if (blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < a.cp_n; j += SEQ_NT) a.cp_dst[j] = a.cp_src[j];
    if (threadIdx.x < SEQ_T) pml[threadIdx.x] = 0.f;

    int tg = tb0;
    int vcur = -1;
    while (tg < tb1) {
        // tiles of this pass: same activation vector
        const int v0 = a.vec_of_q[tg / CB];
        int nt = 1;
        while (nt < SEQ_TB && tg + nt < tb1 && a.vec_of_q[(tg + nt) / CB] == v0) nt++;
        for (int e = threadIdx.x; e < SEQ_TB * 2 * 4 * 64; e += SEQ_NT) (&accl[0][0][0][0])[e] = 0.0;
        if (v0 != vcur && threadIdx.x < a.T) recl[threadIdx.x] = seq_rec(a.part, a.amaxv, v0, threadIdx.x);
        vcur = v0;
        __syncthreads();

        const u32x4 *wt[SEQ_TB];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++) wt[i] = a.bimg + ((size_t)(tg + (i < nt ? i : 0)) * KB) * 64 + lane;
        const u32x4 *img = a.img[v0];
        i32x4 acc[SEQ_TB][2][3];
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int b = 0; b < 3; b++) acc[i][mt][b] = i32x4{0, 0, 0, 0};
        switch (nt) {
        case 1: seq_pass<1, 4>(acc, wt, img, KB, wave, lane); break;
        case 2: seq_pass<2, 3>(acc, wt, img, KB, wave, lane); break;
        case 3: seq_pass<3, 3>(acc, wt, img, KB, wave, lane); break;
        case 4: seq_pass<4, 2>(acc, wt, img, KB, wave, lane); break;
        default: seq_pass<5, 2>(acc, wt, img, KB, wave, lane); break;
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
        // epilogue: D[m][n] with n = lane & 15 (weight row of the tile), m = 4 * (lane >> 4) + reg (row of the chunk)
        for (int e = threadIdx.x; e < nt * 2 * 4 * 64; e += SEQ_NT) {
            const int ln = e & 63, r = (e >> 6) & 3, mt = (e >> 8) & 1, i = e >> 9;
            const int t = mt * 16 + 4 * (ln >> 4) + r;
            const int id = tg + i, q = id / CB, ch = 16 * (id % CB) + (ln & 15);
            const int row = Q * ch + q;
            if (t < a.T && ch < nch && row < N) {
                const SeqRec rc = recl[t];
                const double M = accl[i][mt][r][ln];
                const float v = (float)(rc.scale * (M + rc.cA + SEQ_CU * (double)a.rs[row])) + rc.So;
                a.out[(size_t)t * N + row] = v;
                if (a.epi == 3 && q < 4) {           // what k_seq_stage<1> will stage: relu(k)^2 * r_fv (rwkv.cu:189-190)
                    float h = v * (float)(v > 0.f);
                    h = h * h * a.r_next[4 * ch + q];
                    atomicMax(reinterpret_cast<unsigned *>(&pml[t]), __float_as_uint(fabsf(h)));   // non-negative floats order like their bit patterns
                }
            }
        }
        __syncthreads();
        tg += nt;
    }
    if (a.epi == 3 && threadIdx.x < SEQ_T) a.pmax[(size_t)blockIdx.x * SEQ_T + threadIdx.x] = pml[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
struct SeqGemmTwArgs {
    const u32x4 *bimg;           // B-operand image, Q = 1: tile id = 16-row block
    const u32x4 *img;            // A-operand image of the input vector
    double *part;                // [SEQ_O][SEQ_T][N] exact partial sums
    int N, K, T;
    const double *cp_src;        // piggy-back state commit (see SeqGemmArgs)
    double *cp_dst;
    int cp_n;
};
constexpr int SEQ_TW_CH = 8;     // k-blocks of the activation image staged in LDS at a time (48 KiB)
// "one tile per wave, K over the XCDs": workgroup (i, j), i = 128-row block, j = K-slice = blockIdx % 8 (the XCD the
// workgroup is dispatched to, so that the 32 workgroups of an XCD all read the SAME slice of the activation image through
// their L2); wave w owns row tile 8 i + w for the whole slice; the slice's activation image goes through LDS once for
// all 8 waves, in pieces of SEQ_TW_CH k-blocks (two buffers); the weights stream from the B image, 1 KiB per k-block per
// wave, requested 8 k-blocks ahead.
// TAG only names the launch for the profiler: 0 att_out, 1 ffn_v
template <int TAG>
__global__ __launch_bounds__(SEQ_NT) void k_seq_gemm_tw(SeqGemmTwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *abuf = reinterpret_cast<u32x4 *>(smem);      // [2][SEQ_TW_CH][2][3][64] units
    constexpr int CHU = SEQ_TW_CH * 2 * 3 * 64;         // units per buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, KB = K >> 6, N = a.N;
    const int j = blockIdx.x % SEQ_O, ib = blockIdx.x / SEQ_O;
    const int kb0 = (int)(((long long)j * KB) / SEQ_O), kb1 = (int)(((long long)(j + 1) * KB) / SEQ_O);
    const int ntile = (N + 15) >> 4;
    const int tile = ib * SEQ_NW + wave;
    const bool live = tile < ntile;
    if (blockIdx.x == gridDim.x - 1)
        for (int q = threadIdx.x; q < a.cp_n; q += SEQ_NT) a.cp_dst[q] = a.cp_src[q];

    const u32x4 *wt = a.bimg + ((size_t)(live ? tile : 0) * KB) * 64 + lane;
    constexpr int DW = 8;                                // weight k-blocks in flight per wave
    u32x4 bw[DW];
    const int nkb = kb1 - kb0;
    auto kbc = [&](int it) { return kb0 + (it < nkb ? it : nkb - 1); };
#pragma unroll
    for (int d = 0; d < DW; d++) bw[d] = __builtin_nontemporal_load(wt + (size_t)kbc(d) * 64);
    i32x4 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int b = 0; b < 3; b++) acc[mt][b] = i32x4{0, 0, 0, 0};

    // chunk c of the slice's activation image: units [(kb0 + c CH) * 384, + ncb * 384) are contiguous in the image;
    // 6 units per thread, requested BEFORE the MFMAs of the current chunk and written to the other LDS buffer after them
    constexpr int UPT = SEQ_TW_CH * 384 / SEQ_NT;        // 6
    static_assert(UPT * SEQ_NT == SEQ_TW_CH * 384, "chunk units must divide over the workgroup");
    u32x4 sa[UPT];
    auto stage_load = [&](int c) {
        const int kbs = kb0 + c * SEQ_TW_CH, nu = min(SEQ_TW_CH, kb1 - kbs) * 384;
        const u32x4 *src = a.img + (size_t)kbs * 384;
#pragma unroll
        for (int q = 0; q < UPT; q++) { const int u = threadIdx.x + q * SEQ_NT; sa[q] = src[u < nu ? u : nu - 1]; }
    };
    auto stage_write = [&](int buf) {
        u32x4 *dst = abuf + (size_t)buf * CHU;
#pragma unroll
        for (int q = 0; q < UPT; q++) dst[threadIdx.x + q * SEQ_NT] = sa[q];
    };
    const int nch = (nkb + SEQ_TW_CH - 1) / SEQ_TW_CH;
    if (nch > 0) { stage_load(0); stage_write(0); }
    __syncthreads();
    int it = 0;
    for (int c = 0; c < nch; c++) {
        if (c + 1 < nch) stage_load(c + 1);
        const u32x4 *ab = abuf + (size_t)(c & 1) * CHU + lane;
        const int ncb = min(SEQ_TW_CH, nkb - c * SEQ_TW_CH);
#pragma unroll
        for (int k = 0; k < SEQ_TW_CH; k++) {
            if (k < ncb) {
                const u32x4 w = bw[k % DW];
                bw[k % DW] = __builtin_nontemporal_load(wt + (size_t)kbc(it + DW) * 64);
                const i32x4 bf = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const u32x4 av = ab[((k * 2 + mt) * 3 + b) * 64];
                        const i32x4 af = i32x4{(int)av[0], (int)av[1], (int)av[2], (int)av[3]};
                        acc[mt][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[mt][b], 0, 0, 0);
                    }
                it++;
            }
        }
        if (c + 1 < nch) stage_write((c + 1) & 1);       // the other buffer: its last readers passed the previous barrier
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int t = mt * 16 + 4 * (lane >> 4) + r, row = tile * 16 + (lane & 15);
                if (t < a.T && row < N)
                    a.part[((size_t)j * SEQ_T + t) * N + row] =
                        (double)acc[mt][0][r] + 256.0 * (double)acc[mt][1][r] + 65536.0 * (double)acc[mt][2][r];
            }
    }
}
constexpr size_t SEQ_TW_SMEM = (size_t)2 * SEQ_TW_CH * 2 * 3 * 64 * 16;   // 96 KiB

} // namespace rwkvk
