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
//   * every row of an activation vector is quantised per OCTANT (an eighth of K, whole k-blocks) with the octant's own
//     exact max|.|: no workgroup needs a row-wide maximum, and the scales are finer than one per row;
//   * k_seq_gemm_p / k_seq_gemm_b (K/V/R, att_out, ffn k/r, ffn_v): K-SLICE j = octant j runs on XCD j (blockIdx % 8, so the 32 workgroups
//     of an XCD read the SAME slice of the activation image through their L2 and every workgroup reads an eighth of it);
//     a workgroup = a block of row tiles x one slice, each wave owns NTW row tiles for the whole slice, ALL its weight
//     loads (NTW x k-blocks KiB) are requested up front, the slice's activation image goes through LDS once for all 8 waves;
//     output: per-slice partial values scale_j (M_j + corrections_j) as f32 [8][T][N], added up by the consumer;
//   * k_seq_gemm_ks (head: 50277 rows, partials would be 50 MB): wave w takes octant w of K over the workgroup's <= 5
//     row tiles, per-octant values meet in LDS, epilogue in the kernel;
//   * element-wise work runs on (row, octant) workgroups -- 256 of them for a full chunk: k_seq_resid (partial sums ->
//     residual update + LayerNorm partial statistics), k_seq_site (LayerNorm + token shift + quantisation), k_seq_stage
//     (relu^2 / gated wkv -> quantisation), k_seq_wkv (the recurrence).
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

constexpr int SEQ_T = 32;                 // rows per HALF of a pass: two 16-row MFMA tiles
constexpr int SEQ_TM = 64;                // rows per weight pass (round 4): one or two halves.  A pass of 33..64 rows reads every weight
                                          // byte ONCE for both halves: the GEMMs multiply each weight fragment against both halves' activation
                                          // images (k_seq_gemm_p's NH), everything else sees a global row tg = 32 h + t and finds half h's
                                          // images, records and partial values `*_h` elements behind half 0's
constexpr int SEQ_O = 8;                  // octants of a row (element-wise workgroups per row) = K-slices = XCDs
constexpr double SEQ_CU = 4227200.0;      // 128 * (1 + 256 + 65536) - 2^22: weight-row-sum coefficient
typedef int i32x4 __attribute__((ext_vector_type(4)));

// A-operand image of one activation vector for a chunk: 16-byte units indexed
// [k-block of 64][row tile of 16][limb][lane], lane = 16 * ((k % 64) / 16) + row % 16 -- exactly the
// register image of v_mfma_i32_16x16x64_i8's A operand, so a wave loads a fragment with one
// coalesced 1 KiB global_load_dwordx4.
__device__ __forceinline__ size_t a_unit(int kb, int mt, int limb, int lane) { return (((size_t)kb * 2 + mt) * 3 + limb) * 64 + lane; }
__host__ __device__ inline size_t a_image_bytes(size_t K) { return (K / 64) * 2 * 3 * 64 * 16; }

// quantisation record of one (vector, row, octant): amax -> the octant's fixed-point scale;
// cA = 128 * sum_b 256^b * sum_k (limb_b[k] - 128) over the octant; So = sum_k f[k] * o[k] over the octant
struct SeqPart { double cA; float So; float amax; };
struct SeqStat { double sx, sxx; };                       // per (row, octant): sum x, sum x^2
// offset term of (vector m, row t): the octants' So added up (f32, fixed order)
__device__ __forceinline__ float seq_so(const SeqPart *part, int m, int t)
{
    const SeqPart *p = part + ((size_t)m * SEQ_T + t) * SEQ_O;
    float so = 0.f;
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) so += p[o].So;
    return so;
}
// Per-slice partial values of a GEMM live in the ACCUMULATOR IMAGE of the MFMA tiles, pk[slice][tile id][row tile][reg][lane]
// (so a wave stores a register with one coalesced 256-byte store): element (chunk row t, channel ch of class q) sits in
// tile id = q * CB + ch / 16 at lane 16 * ((t % 16) / 4) + ch % 16, register t % 4, row tile t / 16.
// (-DRWKV_PK_ROWMAJOR=1, round 4: pk[slice][row][tile id][16 channels] instead -- a register row of the image holds FOUR chunk rows,
// so a (row, octant) element-wise workgroup uses a quarter of every 256-byte line it touches, and row-major it reads contiguously.
// Measured and not adopted: k_seq_stage<1> 8.7 -> 8.1 us, but k_seq_wkv<64> -- 16 channels x all rows per workgroup, whole lines of
// the image -- 8.3 -> 12.3 us and the GEMMs' scattered 64-byte stores +0.3-1 us each: 32-token chunk 3.39 -> 3.49 ms, 512-token prompt
// 16.7 -> 16.4 k tokens/s on one box, profiles/r04/pk_rowmajor_ab.txt.)
#ifndef RWKV_PK_ROWMAJOR
#define RWKV_PK_ROWMAJOR 0
#endif
__device__ __forceinline__ size_t pk_index(int ntiles, int slice, int id, int t, int c16)
{
#if RWKV_PK_ROWMAJOR
    return ((((size_t)slice * SEQ_T + t) * ntiles + id) << 4) + c16;
#else
    return ((((size_t)slice * ntiles + id) * 2 + (t >> 4)) * 4 + (t & 3)) * 64 + 16 * ((t & 15) >> 2) + c16;
#endif
}
// where a wave's lane stores accumulator register r of (slice j, tile id, row tile mt): base pointer + r * pk_rstride(ntiles)
__device__ __forceinline__ size_t pk_lane_base(int ntiles, int j, int id, int mt, int lane)
{
#if RWKV_PK_ROWMAJOR
    return ((((size_t)j * SEQ_T + mt * 16 + 4 * (lane >> 4)) * ntiles + id) << 4) + (lane & 15);     // row t = 16 mt + 4 (lane / 16) + r
#else
    return ((((size_t)j * ntiles + id) * 2 + mt) * 4) * 64 + lane;
#endif
}
__device__ __forceinline__ size_t pk_rstride(int ntiles) { return RWKV_PK_ROWMAJOR ? ((size_t)ntiles << 4) : (size_t)64; }
// value of a GEMM output from its per-slice partials (each already scaled and corrected) and the offset term
__device__ __forceinline__ float seq_val(const float *pk, int ntiles, int id, int t, int c16, float so)
{
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < SEQ_O; q++) v += (double)pk[pk_index(ntiles, q, id, t, c16)];
    return (float)v + so;
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
__device__ __forceinline__ void seq_finish(unsigned (&ls)[3], double So, int nk, float amax, SeqPart *dst, double *red)
{
    double s[4] = {(double)ls[0], (double)ls[1], (double)ls[2], So};   // the limb sums of a thread are < 2^32: exact in f64
    eblock_sum<4>(s, red);
    if (threadIdx.x == 0) {
        double ca = 0.0, f = 128.0;
#pragma unroll
        for (int b = 0; b < 3; b++) { ca += f * (s[b] - 128.0 * (double)nk); f *= 256.0; }
        SeqPart p;
        p.cA = ca; p.So = (float)s[3]; p.amax = amax;
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
// (row, octant) workgroups: fold the K-slice partial values of att_out / ffn_v into the residual stream and emit the
// LayerNorm partial statistics of the new x.  MODE 0: statistics only (first layer of a chunk); 1: att_out
// (x = f32(x) + v, rwkv.cu:548-553); 2: ffn_v (x += v * sigmoid(r), :574-577,:407,:212).
struct SeqResidArgs {
    double *x;                   // [T][D]
    const float *pk;             // per-slice partial values of the GEMM (Q = 1: tile id = channel / 16)
    const SeqPart *qpart;        // records of the GEMM's input vector: [T][SEQ_O]
    const float *pk_gate;        // MODE 2: partials of the ffn k/r GEMM (5 classes), r = class 4
    const SeqPart *qpart_gate;   //         records of the ffn r input vector
    SeqStat *stat;               // [T][SEQ_O]
    int D, T;
    size_t pk_h, pkg_h, part_h, partg_h;   // second half of a pass (rows 32..63): element offsets of its pk / pk_gate / qpart / qpart_gate
};
template <int MODE>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_resid(SeqResidArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    // (every pointer with the kernel's FIRST scalar loads: hipcc sinks an argument's s_load to its first use, and every such use behind a wait
    // is one more scalar-cache round trip on a kernel that is a launch and a handful of round trips long -- DESIGN.md 4.3, round 6 for this path)
    RWKV_ARGS_NOW(a.x, a.pk, a.qpart, a.pk_gate, a.qpart_gate, a.stat);
    const int D = a.D, tg = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;      // tg: row of the pass; t: row of its half
    const int hh = tg / SEQ_T, t = tg % SEQ_T;
    const float *pk = a.pk + hh * a.pk_h, *pkg = a.pk_gate + hh * a.pkg_h;
    int c0, c1;
    octant_range(D, o, c0, c1);
    // Every load of the workgroup is requested before the first one is used: the octants' offset terms, then x and the partial values of up to
    // SEQ_RJ channels per thread (D / 8 <= SEQ_RJ * SEQ_ENT).  (Written as "offset terms first, then a loop over the channels" the kernel was a
    // chain of dependent round trips -- the sum of the offset terms in front of the channels' loads -- on a launch that is a few round trips long.)
    constexpr int SEQ_RJ = 3;
    const SeqPart *qp = a.qpart + hh * a.part_h + (size_t)t * SEQ_O, *qg = a.qpart_gate + hh * a.partg_h + (size_t)t * SEQ_O;
    float sov[SEQ_O], sgv[SEQ_O];
#pragma unroll
    for (int q = 0; q < SEQ_O; q++) { sov[q] = MODE != 0 ? qp[q].So : 0.f; sgv[q] = MODE == 2 ? qg[q].So : 0.f; }
    const int CB = (D + 15) >> 4;
    double xs[SEQ_RJ];
    float pv[SEQ_RJ][SEQ_O], pg[SEQ_RJ][SEQ_O];
#pragma unroll
    for (int i = 0; i < SEQ_RJ; i++) {
        const int j = c0 + threadIdx.x + i * SEQ_ENT, jc = j < c1 ? j : (c1 > 0 ? c1 - 1 : 0);      // (an octant of a narrow model may be empty: any valid address)
        xs[i] = a.x[(size_t)tg * D + jc];
#pragma unroll
        for (int q = 0; q < SEQ_O; q++) {
            pv[i][q] = MODE != 0 ? pk[pk_index(CB, q, jc >> 4, t, jc & 15)] : 0.f;
            pg[i][q] = MODE == 2 ? pkg[pk_index(5 * CB, q, 4 * CB + (jc >> 4), t, jc & 15)] : 0.f;
        }
    }
    float so = 0.f, sog = 0.f;          // (seq_so: the octants' So added up in f32, fixed order)
#pragma unroll
    for (int q = 0; q < SEQ_O; q++) { so += sov[q]; sog += sgv[q]; }
    double s[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < SEQ_RJ; i++) {
        const int j = c0 + threadIdx.x + i * SEQ_ENT;
        if (j < c1) {
            double x = xs[i];
            if (MODE != 0) {
                double vd = 0.0, rd = 0.0;      // (seq_val: the slices added up in f64, fixed order)
#pragma unroll
                for (int q = 0; q < SEQ_O; q++) { vd += (double)pv[i][q]; rd += (double)pg[i][q]; }
                const float v = (float)vd + so;
                if (MODE == 1) x = (double)((float)x + v);
                else {
                    const float r = (float)rd + sog;
                    const float gt = (float)(1.0 / (1.0 + exp(-(double)r)));
                    x = x + (double)(v * gt);
                }
                a.x[(size_t)tg * D + j] = x;
            }
            s[0] += x; s[1] += x * x;
        }
    }
    eblock_sum<2>(s, red);
    if (threadIdx.x == 0) {
        SeqStat r;
        r.sx = s[0]; r.sxx = s[1];
        a.stat[tg * SEQ_O + o] = r;
    }
}

// ------------------------------------------------------------------------------------------
struct SeqSiteArgs {
    const double *x;             // [T][D] residual stream
    const SeqStat *stat;         // [T][SEQ_O]
    const double *lnw, *lnb;     // this site's LayerNorm rows
    const double *mix[3];        // token-shift mix per vector (nullptr: no shift, ln_out -> head)
    const float *r[3], *o[3];    // scale / offset of the matrices the vectors feed
    const double *state;         // previous LayerNorm output (state xy / dd of this layer, slot 0): row 0's shift input
    double *state_new;           // GPT: [D] LayerNorm output of the last row (copied over the state afterwards)
    int par;                     // PARRALEL mode (rwkv.cu:236-240): row t is an independent sequence with state slot slot0 + t --
    double *state_par;           //   shift input and state write both go to that slot of this array (same base as `state`)
    size_t slot_stride;          //   L * D
    int slot0;
    unsigned *img[3];            // A-operand images
    SeqPart *part;               // [NV][T][SEQ_O]
    int D, T;
    size_t img_h, part_h;        // second half of a pass: offsets of its images (in 32-bit words) and records
};
// mean, rstd of row t from its octant partials (every thread computes them: 8 tiny loads)
__device__ __forceinline__ void seq_row_stats(const SeqStat *stat, int t, int D, double &mean, double &rstd)
{
    double sx = 0.0, sxx = 0.0;
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) {
        const SeqStat r = stat[t * SEQ_O + o];
        sx += r.sx; sxx += r.sxx;
    }
    mean = sx / (double)D;
    rstd = 1.0 / sqrt((sxx - sx * mean) / (double)(D - 1));    // reference: (D-1), no epsilon (rwkv.cu:43-44,53)
}
__device__ __forceinline__ void seq_row_stats2(const SeqStat *stat, int t, int tp, int D, double &mean, double &rstd, double &meanp, double &rstdp)
{
    SeqStat r[SEQ_O], rp[SEQ_O];
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) { r[o] = stat[t * SEQ_O + o]; rp[o] = stat[tp * SEQ_O + o]; }
    double sx = 0.0, sxx = 0.0, px = 0.0, pxx = 0.0;
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) { sx += r[o].sx; sxx += r[o].sxx; px += rp[o].sx; pxx += rp[o].sxx; }
    mean = sx / (double)D;
    rstd = 1.0 / sqrt((sxx - sx * mean) / (double)(D - 1));
    meanp = px / (double)D;
    rstdp = 1.0 / sqrt((pxx - px * meanp) / (double)(D - 1));
}
// (row, octant) workgroups, one quad (4 channels) per thread: LayerNorm + shift mix of the octant, its exact max|.| per
// vector (one workgroup reduction), quantisation into the A image.  D / 32 <= SEQ_ENT quads per octant.
template <int NV>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_site(SeqSiteArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    RWKV_ARGS_NOW(a.x, a.stat, a.lnw, a.lnb, a.state, a.state_par, a.state_new, a.part);      // (see k_seq_resid)
    RWKV_ARGS_NOW(a.mix[0], a.r[0], a.o[0], a.img[0]);
    if (NV > 1) RWKV_ARGS_NOW(a.mix[NV - 1], a.r[NV - 1], a.o[NV - 1], a.img[NV - 1], a.mix[NV > 2 ? 1 : 0], a.r[NV > 2 ? 1 : 0], a.o[NV > 2 ? 1 : 0], a.img[NV > 2 ? 1 : 0]);
    const int D = a.D, t = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;      // t: row of the pass (0 .. 63)
    const int hh = t / SEQ_T, th = t % SEQ_T;                                 // its half and the row inside the half (images, records)
    int c0, c1;
    octant_range(D, o, c0, c1);
    const bool shift = a.mix[0] != nullptr;
    const bool lnprev = shift && t > 0 && !a.par;   // the shift input is a LayerNorm output: of the previous row (GPT) or already stored (state)
    const double *xprow = !shift ? a.x : a.par ? a.state_par + (size_t)(a.slot0 + t) * a.slot_stride : (t > 0 ? a.x + (size_t)(t - 1) * D : a.state);
    // both rows' statistics in ONE round trip (unconditionally: a branch around the second read is a second trip where it is taken)
    double mean, rstd, meanp, rstdp;
    seq_row_stats2(a.stat, t, lnprev ? t - 1 : t, D, mean, rstd, meanp, rstdp);
    if (!lnprev) { meanp = 0.0; rstdp = 1.0; }
    const int qd = (c0 >> 2) + threadIdx.x;
    const bool live = qd < (c1 >> 2);
    float v[NV][4];
    double So[NV];
    float amax[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { So[m] = 0.0; amax[m] = 0.f; v[m][0] = v[m][1] = v[m][2] = v[m][3] = 0.f; }
    if (live) {
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
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float f = (float)xx[e];
                if (shift) f = (float)(xx[e] * mk[e] + xprev[e] * (1.0 - mk[e]));   // :339-343,:377-384
                v[m][e] = f * rr[e];
                So[m] += (double)(f * oo[e]);
                amax[m] = fmaxf(amax[m], fabsf(v[m][e]));
            }
        }
    }
    eblock_max<NV>(amax, red);
#pragma unroll
    for (int m = 0; m < NV; m++) {
        unsigned ls[3] = {0u, 0u, 0u};
        if (live) seq_store_quad(a.img[m] + hh * a.img_h, qd, th, v[m], inv_scale(amax[m]), ls);
        seq_finish(ls, So[m], c1 - c0, amax[m], a.part + hh * a.part_h + ((size_t)m * SEQ_T + th) * SEQ_O + o, red);
    }
}

// ------------------------------------------------------------------------------------------
struct SeqStageArgs {
    const float *src;            // KIND 0: gated wkv y [T][D]
    const float *pk;             // KIND 1: partials of the ffn k/r GEMM (5 classes): k of hidden unit 4 i + q = class q, channel i
    const SeqPart *qpart_k;      //         records of the ffn k input vector (its offset term)
    const float *r, *o;          // scale / offset over K
    unsigned *img;
    SeqPart *part;               // [T][SEQ_O]
    int K, T;
    size_t pk_h, partk_h, img_h, part_h;     // second half of a pass: offsets of its pk / qpart_k / image (32-bit words) / records
};
constexpr int SEQ_SQ = 3;        // quads per thread of a stage workgroup: K / 32 <= 3 * SEQ_ENT
// (row, octant) workgroups.  KIND 0: f = src (att_out input).  KIND 1: f = relu(k)^2 with k = ffn_k GEMM output
// (rwkv.cu:189-190), K = 4D.  Exact max|f * r| of the octant by one workgroup reduction, then quantisation.
template <int KIND>
__global__ __launch_bounds__(SEQ_ENT) void k_seq_stage(SeqStageArgs a)
{
    __shared__ double red[SEQ_ENW * 4];
    RWKV_ARGS_NOW(a.src, a.pk, a.qpart_k, a.r, a.o, a.img, a.part);      // (see k_seq_resid)
    const int K = a.K, tg = blockIdx.x / SEQ_O, o = blockIdx.x % SEQ_O;
    const int hh = tg / SEQ_T, t = tg % SEQ_T;
    const float *pk = a.pk + hh * a.pk_h;
    int c0, c1;
    octant_range(K, o, c0, c1);
    const float sok = KIND == 1 ? seq_so(a.qpart_k + hh * a.partk_h, 0, t) : 0.f;
    const int q0 = c0 >> 2, q1 = c1 >> 2;
    float v[SEQ_SQ][4];
    double So = 0.0;
    float am[1] = {0.f};
#pragma unroll
    for (int i = 0; i < SEQ_SQ; i++) {
        const int qd = q0 + threadIdx.x + i * SEQ_ENT;
        v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        if (qd < q1) {
            float f[4];
            if (KIND == 0) {
                const f32x4 sv = reinterpret_cast<const f32x4 *>(a.src + (size_t)tg * K)[qd];
                f[0] = sv[0]; f[1] = sv[1]; f[2] = sv[2]; f[3] = sv[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int CB = ((K >> 2) + 15) >> 4;
                    float k = seq_val(pk, 5 * CB, e * CB + (qd >> 4), t, qd & 15, sok);   // hidden units 4 qd .. 4 qd + 3 = classes 0..3 of channel qd
                    k = k * (float)(k > 0.f);
                    f[e] = k * k;                                                          // relu(k)^2, rwkv.cu:189-190
                }
            }
            const f32x4 rv = reinterpret_cast<const f32x4 *>(a.r)[qd], ov = reinterpret_cast<const f32x4 *>(a.o)[qd];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[i][e] = f[e] * rv[e];
                So += (double)(f[e] * ov[e]);
                am[0] = fmaxf(am[0], fabsf(v[i][e]));
            }
        }
    }
    eblock_max<1>(am, red);
    const float inv_s = inv_scale(am[0]);
    unsigned ls[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < SEQ_SQ; i++) {
        const int qd = q0 + threadIdx.x + i * SEQ_ENT;
        if (qd < q1) seq_store_quad(a.img + hh * a.img_h, qd, t, v[i], inv_s, ls);
    }
    seq_finish(ls, So, c1 - c0, am[0], a.part + hh * a.part_h + (size_t)t * SEQ_O + o, red);
}

// ------------------------------------------------------------------------------------------
struct SeqWkvArgs {
    const float *pk;             // partials of the K/V/R GEMM (3 classes)
    const SeqPart *qpart;        // records of the three input vectors [3][T][SEQ_O]
    const double *uw, *ew;       // bonus + decay, exp(decay) of this layer
    double *saa, *sbb;           // state of this layer, slot 0
    float *y;                    // [T][D] gated wkv, cast to f32 as the att_out GEMV does (rwkv.cu:290)
    int D, T;
    int par;                     // PARRALEL mode: row t uses state slot slot0 + t (no recurrence along the rows)
    size_t slot_stride;
    int slot0;
    size_t pk_h, part_h;         // second half of a pass: offsets of its pk / records
};
constexpr int WKV_CH = 16;       // channels per workgroup (TCAP x 16 threads: 16 channels x the rows of the pass)
// rwkv.cu:242-255 with the GPT-mode state slot 0.  The exponentials do not depend on the state, so
// one thread per (row, channel) evaluates them; then one thread per channel runs the (linear) state
// recurrence along the chunk and the outputs are finished in parallel.  TCAP = 32 (one half) or 64 rows.
template <int TCAP>
__global__ __launch_bounds__(TCAP * WKV_CH) void k_seq_wkv(SeqWkvArgs a)
{
    __shared__ double e1s[TCAP][WKV_CH], eks[TCAP][WKV_CH], vs[TCAP][WKV_CH], sgs[TCAP][WKV_CH], aas[TCAP][WKV_CH];
    RWKV_ARGS_NOW(a.pk, a.qpart, a.uw, a.ew, a.saa, a.sbb, a.y);      // (see k_seq_resid)
    const int ch = threadIdx.x & (WKV_CH - 1), t = threadIdx.x / WKV_CH;     // t: row of the pass
    const int i = blockIdx.x * WKV_CH + ch;
    const bool live = i < a.D && t < a.T;
    if (live) {
        const int CB = (a.D + 15) >> 4, hh = t / SEQ_T, th = t % SEQ_T;
        const float *pk = a.pk + hh * a.pk_h;
        const SeqPart *qp = a.qpart + hh * a.part_h;
        const float k = seq_val(pk, 3 * CB, 0 * CB + (i >> 4), th, i & 15, seq_so(qp, 0, th));
        const float v = seq_val(pk, 3 * CB, 1 * CB + (i >> 4), th, i & 15, seq_so(qp, 1, th));
        const float r = seq_val(pk, 3 * CB, 2 * CB + (i >> 4), th, i & 15, seq_so(qp, 2, th));
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
            a.y[(size_t)t * a.D + i] = (float)(sgs[t][ch] * ((aa + e1 * vv) / (bb + e1)));
            a.saa[so] = (aa + ek * vv) * ew;
            a.sbb[so] = (bb + ek) * ew;
        }
        return;
    }
    // GPT mode: the state recurrences aa' = (aa + e^k v) e^w, bb' = (bb + e^k) e^w are LINEAR -- one thread per channel walks
    // them along the chunk (two fma per step) and leaves the state every row sees; the divisions of all (row, channel) pairs
    // then run in parallel (rwkv.cu:242-255 evaluates the same expressions in the same order per row)
    if (threadIdx.x < WKV_CH && i < a.D) {
        double aa = a.saa[i], bb = a.sbb[i];
        const double ew = a.ew[i];
        for (int tt = 0; tt < a.T; tt++) {
            const double ek = eks[tt][ch], vv = vs[tt][ch];
            eks[tt][ch] = aa;                 // state before row tt (the slot of e^k is free once read)
            aas[tt][ch] = bb;
            aa = (aa + ek * vv) * ew;
            bb = (bb + ek) * ew;
        }
        a.saa[i] = aa; a.sbb[i] = bb;
    }
    __syncthreads();
    if (live) {
        const double e1 = e1s[t][ch];
        a.y[(size_t)t * a.D + i] = (float)(sgs[t][ch] * ((eks[t][ch] + e1 * vs[t][ch]) / (aas[t][ch] + e1)));
    }
}

// ------------------------------------------------------------------------------------------
// Load-time: tile image of a re-tiled matrix w_t[N][K] (row-per-output).  A tile is TH rows (16: the MFMA B operand of the chunk path, also
// what the tile-form decode kernels of a 4096-wide model stream; 4: the decode-only image of widths whose channels do not split into
// 16-row blocks per workgroup, tile.hip.h); a fragment (1 KiB, one wave instruction) holds TH rows x 64 / TH 16-byte pieces of k.  Tiles
// are enumerated class-major: tile id -> class q = id / CB, TH-channel block cb = id % CB, tile row c -> matrix row Q * (TH cb + c) + q
// (Q row classes interleaved in w_t: K/V/R 3, ffn k,k,k,k,r 5, else 1), so that consecutive tiles share an activation vector.
// bimg[((id * KB + kb) * 64 + lane) * 16 + b] = w_t[row(id, lane % TH)][(1024 / TH) kb + 16 (lane / TH) + b] - 128 (rows past N: 0); KB = K TH / 1024.
__global__ void k_bimage(const uint8_t *__restrict__ w_t, uint8_t *__restrict__ bimg, int N, int K, int Q, int CB, int TH)
{
    const int KF = 1024 / TH, KB = K / KF;
    const size_t unit = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte unit per thread
    const size_t total = (size_t)Q * CB * KB * 64;
    if (unit >= total) return;
    const int lane = (int)(unit & 63);
    const size_t tk = unit >> 6;
    const int kb = (int)(tk % KB), id = (int)(tk / KB);
    const int q = id / CB, cb = id % CB;
    const int ch = TH * cb + (lane % TH), row = Q * ch + q;
    const int nch = (N + Q - 1) / Q;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (ch < nch && row < N) {
        v = *reinterpret_cast<const u32x4 *>(w_t + (size_t)row * K + (size_t)kb * KF + 16 * (lane / TH));
#pragma unroll
        for (int d = 0; d < 4; d++) v[d] ^= 0x80808080u;
    }
    reinterpret_cast<u32x4 *>(bimg)[unit] = v;
}
// Load-time: row sums of the unsigned weights per OCTANT of K: rs8[o][row] = sum over k in octant o of w_t[row][k].  One wave per row.
__global__ void k_rowsum8(const uint8_t *__restrict__ w_t, unsigned *__restrict__ rs8, int N, int K)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63, KB = K >> 6;
    unsigned acc[SEQ_O];
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) acc[o] = 0u;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(w_t + (size_t)row * K);
    for (int c = lane; c < (K >> 4); c += 64) {
        const u32x4 v = p[c];
        unsigned sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) sum = __builtin_amdgcn_udot4(v[q], 0x01010101u, sum, false);
        const int kb = c >> 2;
#pragma unroll
        for (int o = 0; o < SEQ_O; o++) {
            const int k0 = (int)(((long long)o * KB) / SEQ_O), k1 = (int)(((long long)(o + 1) * KB) / SEQ_O);
            acc[o] += (kb >= k0 && kb < k1) ? sum : 0u;
        }
    }
#pragma unroll
    for (int o = 0; o < SEQ_O; o++) {
        const unsigned tot = wave_sum_dpp(acc[o]);
        if (lane == 0) rs8[(size_t)o * N + row] = tot;
    }
}

// ------------------------------------------------------------------------------------------
struct SeqGemmArgs {
    const u32x4 *bimg;           // B-operand image of the weights (k_bimage)
    const unsigned *rs8;         // [SEQ_O][N] row sums of the unsigned weights per octant
    int N, K;
    int Q;                       // row classes interleaved in the matrix
    int vec_of_q[5];             // activation vector each class multiplies (non-decreasing in q)
    const u32x4 *img[3];         // A-operand images of the vectors
    const SeqPart *part;         // quantisation records [NV][T][SEQ_O]
    float *pk;                   // k_seq_gemm_p / _b: per-slice partial values, accumulator image (pk_index)
    float *out;                  // k_seq_gemm_ks: [T][N]
    int T;
    int ntw;                     // k_seq_gemm_p: row tiles per wave in use (<= the template's NTW)
    const double *cp_src;        // piggy-back copy (stream-ordered behind the site kernel that produced it): the chunk's
    double *cp_dst;              // last LayerNorm output -> recurrent state; cp_n == 0: none
    int cp_n;
    size_t img_h, part_h, pk_h;  // second half of a pass (k_seq_gemm_p with NH = 2; k_seq_gemm_ks is launched per half): offsets of its images
                                 // (16-byte units), records and partial values
    unsigned long long *tl;      // optional phase timeline (tl_stamp; tools/gemm_timeline.py): 0 entry, 1 requests issued, 2 activation image
                                 // staged, 3 first batch / chunk multiplied and emitted, 4 last weights multiplied, 5 end
};
constexpr int SEQ_NT = 512;      // GEMM workgroup: 8 waves, two per SIMD
// dynamic LDS of the GEMM kernels
constexpr size_t seq_gemm_p_smem(int nkb, int nvs, bool multi, int nh = 1) { return (size_t)(multi ? 2 : 1) * nh * nvs * nkb * 384 * 16 + (size_t)nh * nvs * SEQ_T * 16; }
constexpr int SEQ_TB = 5;        // weight tiles (16 rows each) per workgroup pass of k_seq_gemm_ks
constexpr size_t SEQ_KS_SMEM = sizeof(float) * SEQ_O * SEQ_TB * 2 * 4 * 64 + sizeof(float) * SEQ_T;
constexpr int SEQ_NW = SEQ_NT / 64;

// value of one slice: scale_o (M + cA_o + CU * rowsum_o)
__device__ __forceinline__ double seq_slice_value(const SeqPart &rc, double M, unsigned rs)
{
    return scale_of(rc.amax) * (M + rc.cA + SEQ_CU * (double)rs);
}

// k_seq_gemm_p -- "tile per wave, K over the XCDs": workgroup (rb, j): j = blockIdx % 8 = K-slice = octant (and the XCD the workgroup is
// dispatched to), rb = block of 8 * ntw row tiles; wave w owns tiles (rb * 8 + w) * ntw + i for the whole slice.  NTW: tiles per wave
// (registers), NKB: k-blocks of the slice's activation image resident in one LDS buffer, NVS: activation vectors the tiles of one
// workgroup may span (LDS).  The K loop is a SOFTWARE PIPELINE over the slice's k-blocks (round 3; round 2's form requested a wave's whole
// weight stream up front: a wave that does so BLOCKS AT ISSUE -- a CU's memory path holds ~16 KB of requests --, the barrier behind
// the requests was reached when three quarters of the stream had arrived and the MFMAs, the epilogue and the partial-value stores ran
// BEHIND the stream instead of under it: 3.3 TB/s, tools/gemm_timeline.py, profiles/r03/gemm_timeline.txt).  A wave keeps DEPTH
// k-blocks of its NTW tiles in flight in a rolling register buffer: step f requests block f + DEPTH - 1, waits for block f, multiplies
// it against both row tiles of the chunk.  Slices longer than NKB k-blocks (MULTI; ffn_v: K = 4 D) re-stage the activation image per
// NKB blocks into the other LDS buffer, requested at the head of the previous group of blocks.
// The weight loads are inline asm like the DMA: hipcc does not count asm in vmcnt, so its own waitcnt insertion would make every
// wait for a weight block drain the DMA pieces requested behind it; all waits for them are explicit (in-order completion).
__device__ __forceinline__ u32x4 load_b_asm(const u32x4 *p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// wait until only the `newer` k-blocks (NTW loads each) requested behind the wanted one -- and, with `dma`, PW DMA pieces -- are outstanding
template <int NTW, int PW, int B> struct WaitBlocks {
    static __device__ __forceinline__ void run(int newer, bool dma)
    {
        if (newer >= B) { if (dma) wait_vm<B * NTW + PW>(); else wait_vm<B * NTW>(); }
        else WaitBlocks<NTW, PW, B - 1>::run(newer, dma);
    }
};
template <int NTW, int PW> struct WaitBlocks<NTW, PW, 0> {
    static __device__ __forceinline__ void run(int, bool dma) { if (dma) wait_vm<PW>(); else wait_vm<0>(); }
};
// NH = 2 (round 4): the pass has two halves of up to 32 rows; every weight fragment is multiplied against BOTH halves' activation
// images (LDS: [half][vector][k][row tile][limb][lane]; twice the accumulators), i.e. the weights are read once per 64 rows.
template <int TAG, int NTW, int NKB, int NVS, int DEPTH, bool MULTI, int NH = 1>
__global__ __launch_bounds__(SEQ_NT) void k_seq_gemm_p(SeqGemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(NKB % DEPTH == 0, "the rolling buffer's slot of a k-block must be a compile-time value");
    constexpr int NBUF = MULTI ? 2 : 1;
    constexpr int CHU = NH * NVS * NKB * 384;           // units of one LDS buffer: [half][vector][k][row tile][limb][lane]
    constexpr int PW = (NH * NVS * NKB * 6 + SEQ_NW - 1) / SEQ_NW;      // DMA pieces (1 KiB) per wave and chunk; the last round is padded with duplicates
    static_assert(PW + DEPTH * NTW + 2 + NTW <= 63, "k_seq_gemm_p: more than 63 vector memory operations in flight");
    u32x4 *abuf = reinterpret_cast<u32x4 *>(smem);
    double *recl = reinterpret_cast<double *>(smem + (size_t)NBUF * CHU * 16);   // [NH][NVS][SEQ_T]{scale, cA} of this slice
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // every pointer and the class -> vector table with the kernel's FIRST scalar loads (round 6: the ISA of this kernel's head was a chain of ten
    // dependent scalar-cache round trips in front of the first request -- arguments fetched where they are first used, and `vec_of_q[dynamic index]`
    // as one s_load per lookup behind the division that makes the index)
    RWKV_ARGS_NOW(a.bimg, a.rs8, a.img[0], a.img[NVS > 1 ? 1 : 0], a.img[NVS > 2 ? 2 : 0], a.part, a.pk, a.tl);
    const int vq0 = a.vec_of_q[0], vq1 = a.vec_of_q[1], vq2 = a.vec_of_q[2], vq3 = a.vec_of_q[3], vq4 = a.vec_of_q[4];
    RWKV_ARGS_NOW(vq0, vq1, vq2, vq3, vq4, a.N, a.K, a.Q);
    auto vec_of = [&](int q) { return q <= 0 ? vq0 : q == 1 ? vq1 : q == 2 ? vq2 : q == 3 ? vq3 : vq4; };
    const u32x4 *const im0 = a.img[0], *const im1 = a.img[1], *const im2 = a.img[2];
    auto img_of = [&](int v) { return v <= 0 ? im0 : v == 1 ? im1 : im2; };      // (a.img[dynamic index] is one scalar load per DMA piece)
    tl_stamp(a.tl, 0);
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nch = (N + Q - 1) / Q, CB = (nch + 15) >> 4, ntiles = Q * CB;
    const int j = blockIdx.x % SEQ_O, rb = blockIdx.x / SEQ_O;
    const int kb0 = (int)(((long long)j * KB) / SEQ_O), kb1 = (int)(((long long)(j + 1) * KB) / SEQ_O);
    const int nkb = kb1 - kb0, nchunk = (nkb + NKB - 1) / NKB;
    if (blockIdx.x == gridDim.x - 1)
        for (int q = threadIdx.x; q < a.cp_n; q += SEQ_NT) a.cp_dst[q] = a.cp_src[q];
    const int ntw = a.ntw;
    const int id0 = (rb * SEQ_NW + wave) * ntw;          // this wave's tiles: id0 .. id0 + ntw - 1
    const int wg0 = rb * SEQ_NW * ntw, wg1 = min(wg0 + SEQ_NW * ntw, ntiles) - 1;
    const int vlo = vec_of(min(wg0, ntiles - 1) / CB), vhi = vec_of(max(wg1, 0) / CB);
    SeqPart rc;
    {
        const int tr = threadIdx.x < NH * NVS * SEQ_T ? (int)threadIdx.x : 0;
        const int hh = tr / (NVS * SEQ_T), v = min(vlo + (tr / SEQ_T) % NVS, vhi), t = tr % SEQ_T;
        rc = a.part[hh * a.part_h + ((size_t)v * SEQ_T + t) * SEQ_O + j];
    }
    const u32x4 *wt[NTW];
    int vi[NTW];
    bool tv[NTW];
    unsigned rsv[NTW];
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        const int id = id0 + i;
        tv[i] = i < ntw && id < ntiles;
        const int idc = tv[i] ? id : 0;
        {
            const int q = idc / CB, ch = 16 * (idc % CB) + (lane & 15), row = Q * ch + q;
            rsv[i] = a.rs8[(size_t)j * N + ((ch < nch && row < N) ? row : 0)];
        }
        wt[i] = a.bimg + ((size_t)idc * KB) * 64 + lane;
        vi[i] = vec_of(idc / CB) - vlo;
        vi[i] = vi[i] < 0 ? 0 : (vi[i] >= NVS ? NVS - 1 : vi[i]);
        vi[i] = __builtin_amdgcn_readfirstlane(vi[i]);
    }
    // activation image of chunk c (k-blocks kb0 + c NKB ...), vectors vlo .. vhi -> LDS buffer `buf`, flat piece index over
    // [vector][1 KiB piece].  A kernel that re-stages (MULTI) must know how many pieces a wave has in flight: every wave issues
    // exactly PW, an index past the end copies piece 0 of vector vlo again (same bytes, same place); the others issue what is needed.
    // (What the image costs: a CU takes in ~25 KB/us whatever the source, and 32 rows x 3 limbs of the slice are 48 KiB per vector
    // at D = 4096 -- 15 % on top of the weights for ffn k/r, 25 % for K/V/R, 75 % for att_out and ffn_v, whose workgroups own 128
    // weight rows only.  The first version of this kernel padded EVERY kind to PW pieces: K/V/R got slower than the kernel it replaces.)
    const unsigned abuf_lds = lds_addr(abuf);
    const int nvec = min(vhi - vlo + 1, NVS);
    auto piece = [&](int p, int kbs, int np, int buf) {
        int hv = p / (NKB * 6), pc = p % (NKB * 6);
        int hh = hv / NVS, v = hv % NVS;
        if (hh >= NH || v >= nvec || pc >= np) { hh = 0; v = 0; pc = 0; }
        const uint8_t *src = reinterpret_cast<const uint8_t *>(img_of(vlo + v) + hh * a.img_h + (size_t)kbs * 384) + lane * 16 + (size_t)pc * 1024;
        const unsigned dst = abuf_lds + (unsigned)(((size_t)buf * CHU + (size_t)(hh * NVS + v) * NKB * 384) * 16) + (unsigned)pc * 1024u;
        dma_piece_shared(np > 0 ? src : reinterpret_cast<const uint8_t *>(img_of(vlo)) + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
    };
    auto stage_a = [&](int c, int buf) {
        const int kbs = kb0 + c * NKB, n = max(min(NKB, nkb - c * NKB), 0);
        const int np = n * 6;
        if (MULTI) {
#pragma unroll
            for (int r = 0; r < PW; r++) piece(wave + r * SEQ_NW, kbs, np, buf);
        } else {
            for (int hv = 0; hv < NH * NVS; hv++)
                if (hv % NVS < nvec)
                    for (int pc = wave; pc < np; pc += SEQ_NW) piece(hv * NKB * 6 + pc, kbs, np, buf);
        }
    };
    u32x4 bwr[DEPTH][NTW];
    // weights of the slice's k-block f into slot f % DEPTH; nothing is requested past the slice's end (a request nobody uses would
    // still have to land before the epilogue: 1-2 us), so the waits count the blocks that really are behind the wanted one
    auto load_k = [&](int f, int slot) {
        if (f < nkb) {
            const int kb = min(kb0 + f, KB - 1);
#pragma unroll
            for (int i = 0; i < NTW; i++) bwr[slot][i] = load_b_asm(wt[i] + (size_t)kb * 64);
        }
    };
    i32x4 acc[NH][NTW][2][3];
#pragma unroll
    for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int i = 0; i < NTW; i++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int b = 0; b < 3; b++) acc[hh][i][mt][b] = i32x4{0, 0, 0, 0};

    stage_a(0, 0);
#pragma unroll
    for (int f = 0; f < DEPTH - 1; f++) load_k(f, f);
    tl_stamp(a.tl, 1);
    WaitBlocks<NTW, PW, DEPTH - 1>::run(min(DEPTH - 1, nkb), false);     // this wave's share of the image is older than the weights
    __syncthreads();
    tl_stamp(a.tl, 2);
    for (int c = 0; c < nchunk; c++) {
        const bool more = MULTI && c + 1 < nchunk;
        const int n = min(NKB, nkb - c * NKB);
        const u32x4 *ab = abuf + (size_t)(MULTI ? (c & 1) : 0) * CHU + lane;
#pragma unroll
        for (int k = 0; k < NKB; k++) {
            if (k >= n) continue;                   // (wave-uniform) past the slice's end
            const int f = c * NKB + k;
            // the re-staging DMA is requested IN FRONT of this step's weight block: the wait for that block (step k = DEPTH - 1, at
            // the latest the chunk's last step) then proves that this wave's pieces have landed -- loads complete in order --, so
            // every wave reaches the end-of-chunk barrier with its share of the next image in LDS (requested behind
            // the block, the pieces were never waited for when DEPTH == NKB)
            if (k == 0 && more) stage_a(c + 1, (c + 1) & 1);
            load_k(f + DEPTH - 1, (k + DEPTH - 1) % DEPTH);
            // block f has landed when only what was requested behind it is outstanding: up to DEPTH - 1 blocks, and -- for the first
            // DEPTH - 1 steps behind a re-staging -- the DMA pieces
            WaitBlocks<NTW, PW, DEPTH - 1>::run(min(DEPTH - 1, nkb - 1 - f), more && k < DEPTH - 1);
            // load_b_asm hands out its destination before the data is there (hipcc does not track asm loads): tell the compiler that
            // block f's registers CHANGE here, so that no read of them can be scheduled above the wait.  (A copy of a register made
            // between the request and this point would still be wrong; none is made -- DEPTH and the slot are compile-time values, the
            // parity gates run for every RWKV_SEQ_DEPTH knob.)
#pragma unroll
            for (int i = 0; i < NTW; i++) asm volatile("" : "+v"(bwr[k % DEPTH][i]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < NH; hh++) {
                u32x4 av[2][3];
                auto read_a = [&](int vv) {
#pragma unroll
                    for (int ms = 0; ms < 2; ms++)
#pragma unroll
                        for (int b = 0; b < 3; b++) av[ms][b] = ab[((((size_t)hh * NVS + vv) * NKB + k) * 2 + ms) * 3 * 64 + b * 64];
                };
                read_a(vi[0]);
#pragma unroll
                for (int i = 0; i < NTW; i++) {
                    if (NVS > 1 && i > 0 && vi[i] != vi[i - 1]) read_a(vi[i]);
                    const u32x4 w = bwr[k % DEPTH][i];
                    const i32x4 bf = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
#pragma unroll
                    for (int ms = 0; ms < 2; ms++)
#pragma unroll
                        for (int b = 0; b < 3; b++) {
                            const i32x4 af = i32x4{(int)av[ms][b][0], (int)av[ms][b][1], (int)av[ms][b][2], (int)av[ms][b][3]};
                            acc[hh][i][ms][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[hh][i][ms][b], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (c == 0) tl_stamp(a.tl, 3);
        if (more) {
            // Every wave's pieces of chunk c + 1 have LANDED here (see the request order above; the restated wait is satisfied
            // already and costs nothing -- as a wait for pieces requested BEHIND the block it cost the ffn_v GEMM 0.9 us per launch);
            // behind the barrier everyone has left chunk c's buffer and may read the other one
            WaitBlocks<NTW, PW, DEPTH - 1>::run(min(DEPTH - 1, nkb - (c + 1) * NKB), false);
            __syncthreads();
        }
    }
    tl_stamp(a.tl, 4);
    // (the record and the row sums are first TOUCHED here: hipcc's own wait for them -- it cannot see the asm requests behind them --
    // would otherwise drain the pipeline in front of the loop)
    if (threadIdx.x < NH * NVS * SEQ_T) { recl[2 * threadIdx.x] = scale_of(rc.amax); recl[2 * threadIdx.x + 1] = rc.cA; }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                if (!tv[i]) continue;
                const int id = id0 + i;
                const double *rl = recl + 2 * (((size_t)hh * NVS + vi[i]) * SEQ_T + mt * 16 + 4 * (lane >> 4));
                float *dst = a.pk + hh * a.pk_h + pk_lane_base(ntiles, j, id, mt, lane);
                const size_t rst = pk_rstride(ntiles);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double M = (double)acc[hh][i][mt][0][r] + 256.0 * (double)acc[hh][i][mt][1][r] + 65536.0 * (double)acc[hh][i][mt][2][r];
                    dst[r * rst] = (float)(rl[2 * r] * (M + rl[2 * r + 1] + SEQ_CU * (double)rsv[i]));   // scale_o (M + cA_o + CU rowsum_o)
                }
            }
    tl_stamp(a.tl, 5);
}

// ------------------------------------------------------------------------------------------
// k_seq_gemm_b (round 4): a 64-row pass's GEMM with ONE vector's activation image of the whole slice resident in LDS, staged once per
// workgroup, and a wave's row tiles multiplied in BATCHES of at most NTW (the accumulators of two halves).
// k_seq_gemm_p<.., MULTI, 2> holds three tiles per wave at 64 rows: ffn k/r then needs 432 workgroups on 256 CUs -- two rounds, 33 us --
// and re-stages the image per two k-blocks.  What a workgroup of these GEMMs takes is launch + the image's first touch (4.6 us during
// which HBM idles), its weights at the CU's share of the stream (22.8 KB/us), an epilogue (2.3 us, idle again): one round of longer
// workgroups beats two rounds (tools/gemm_timeline.py, profiles/r04/gemm_b_timeline.txt).
//  * Workgroup ranges are aligned to the VECTOR GROUPS of the matrix (runs of row classes that multiply the same vector: k|r for ffn
//    k/r) -- a first version with plain ranges ran the two workgroups per slice that straddle a group as one pass per vector, each as long
//    as a whole workgroup's (a wave's stream goes at the loaded memory system's latency whoever else is idle): 26 us where the others took 16.
//    The host gives every group its share of the 32 workgroups of a slice (grp_rb); within a group the tiles are spread evenly over the
//    waves (5 or 6 each for ffn k/r at D = 4096).
//  * The k-blocks of all batches are one stream through the rolling register buffer: the next batch's first blocks (and its row sums)
//    are in flight under the epilogue of the previous one.  Requests per slot are counted at run time, the waits go through wait_vm_dyn.
// Same arithmetic in the same order per output as k_seq_gemm_p: bit-identical partial values.
__device__ __forceinline__ unsigned load_u32_asm(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MAXN> __device__ __forceinline__ void wait_vm_dyn(int n)
{
    if constexpr (MAXN == 0) wait_vm<0>();
    else { if (n >= MAXN) wait_vm<MAXN>(); else wait_vm_dyn<MAXN - 1>(n); }
}
#ifndef RWKV_SEQ_BDEPTH
#define RWKV_SEQ_BDEPTH 3
#endif
constexpr int SEQ_B_NKB_MAX = 10;        // longest slice whose two-half image fits: 2 x 10 x 6 KiB = 120 KiB (D = 5120)
constexpr size_t seq_gemm_b_smem(int nkbm, int nh) { return (size_t)nh * nkbm * 384 * 16 + (size_t)nh * SEQ_T * 16; }
struct SeqGemmBArgs {
    SeqGemmArgs g;
    int grp_tile[4];             // first tile of vector group i (i < ngrp), ntiles behind the last
    int grp_rb[4];               // first workgroup (per slice) of group i, the slice's workgroup count behind the last
    int ngrp;
};
template <int TAG, int NTW, int DEPTH, int NH>
__global__ __launch_bounds__(SEQ_NT) void k_seq_gemm_b(SeqGemmBArgs ba)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert((DEPTH - 1) * 2 * NTW <= 40, "k_seq_gemm_b: the wait dispatch covers 40 requests");
    const SeqGemmArgs &a = ba.g;
    RWKV_ARGS_NOW(a.bimg, a.rs8, a.img[0], a.img[1], a.img[2], a.part, a.pk, a.tl);      // (see k_seq_gemm_p)
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nkbm = (KB + SEQ_O - 1) / SEQ_O;                    // longest slice
    u32x4 *abuf = reinterpret_cast<u32x4 *>(smem);
    double *recl = reinterpret_cast<double *>(smem + (size_t)NH * nkbm * 384 * 16);   // [NH][SEQ_T]{scale, cA} of this slice
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    tl_stamp(a.tl, 0);
    const int nch = (N + Q - 1) / Q, CB = (nch + 15) >> 4, ntiles = Q * CB;
    const int j = blockIdx.x % SEQ_O, rb = blockIdx.x / SEQ_O;
    const int kb0 = (int)(((long long)j * KB) / SEQ_O), kb1 = (int)(((long long)(j + 1) * KB) / SEQ_O);
    const int nkb = kb1 - kb0;
    if (blockIdx.x == gridDim.x - 1)
        for (int q = threadIdx.x; q < a.cp_n; q += SEQ_NT) a.cp_dst[q] = a.cp_src[q];
    // this workgroup's vector group, this wave's tiles id0 .. id0 + tpw - 1 of it
    int gi = 0;
    while (gi + 1 < ba.ngrp && rb >= ba.grp_rb[gi + 1]) gi++;
    const int gt0 = ba.grp_tile[gi], gT = ba.grp_tile[gi + 1] - gt0;
    const int gw = (ba.grp_rb[gi + 1] - ba.grp_rb[gi]) * SEQ_NW, widx = (rb - ba.grp_rb[gi]) * SEQ_NW + wave;
    const int id0 = gt0 + (int)(((long long)widx * gT) / gw), tpw = gt0 + (int)(((long long)(widx + 1) * gT) / gw) - id0;
    const int v = a.vec_of_q[min(gt0, ntiles - 1) / CB];
    const int nb = (tpw + NTW - 1) / NTW, bs = nb ? (tpw + nb - 1) / nb : 0;     // batches (the host keeps them <= 3), tiles per batch (<= NTW)
    SeqPart rc;
    {
        const int tr = threadIdx.x < NH * SEQ_T ? (int)threadIdx.x : 0;
        const int hh = tr / SEQ_T, t = tr % SEQ_T;
        rc = a.part[hh * a.part_h + ((size_t)v * SEQ_T + t) * SEQ_O + j];
    }
    const unsigned abuf_lds = lds_addr(abuf);
    const u32x4 *const wbase = a.bimg + lane;
    auto batch_mask = [&](int b) {       // bit i <-> tile id0 + b * bs + i
        int m = 0;
#pragma unroll
        for (int i = 0; i < NTW; i++)
            if (i < bs && b * bs + i < tpw) m |= 1 << i;
        return m;
    };
    u32x4 bwr[DEPTH][NTW];
    unsigned rsn[NTW], rsv[NTW];          // row sums (this octant) of the rows this lane finishes: of the batch being requested / multiplied
    int cnt[DEPTH];                       // requests issued into slot s (weights + row sums): what is "newer" than a slot's block
    i32x4 acc[NH][NTW][2][3];
    auto zero_acc = [&]() {
#pragma unroll
        for (int hh = 0; hh < NH; hh++)
#pragma unroll
            for (int i = 0; i < NTW; i++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int b = 0; b < 3; b++) acc[hh][i][mt][b] = i32x4{0, 0, 0, 0};
    };
    // the slice's image of vector v, both halves: 1 KiB pieces round-robin over the waves
    {
        const int np = nkb * 6;
        for (int hh = 0; hh < NH; hh++)
            for (int pc = wave; pc < np; pc += SEQ_NW) {
                const uint8_t *src = reinterpret_cast<const uint8_t *>(a.img[v] + hh * a.img_h + (size_t)kb0 * 384) + lane * 16 + (size_t)pc * 1024;
                const unsigned dst = abuf_lds + (unsigned)(hh * nkbm * 384 * 16) + (unsigned)pc * 1024u;
                dma_piece_shared(src, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
            }
    }
    // this wave's stream of (batch, k-block) steps
    const int total = nb * nkb;
    int lb = 0, lf = 0;                                           // request cursor: batch, k-block
    auto request = [&](int slot) {                                // the next step's weights (and, at a batch's first block, its row sums) into `slot`
        int n = 0;
        if (lb < nb) {
            const int m = batch_mask(lb);
            if (lf == 0) {
#pragma unroll
                for (int i = 0; i < NTW; i++)
                    if (m >> i & 1) {
                        const int id = id0 + lb * bs + i;
                        const int q = id / CB, ch = 16 * (id % CB) + (lane & 15), row = Q * ch + q;
                        rsn[i] = load_u32_asm(a.rs8 + (size_t)j * N + ((ch < nch && row < N) ? row : 0));
                        n++;
                    }
            }
#pragma unroll
            for (int i = 0; i < NTW; i++)
                if (m >> i & 1) {
                    const int id = id0 + lb * bs + i;
                    bwr[slot][i] = load_b_asm(wbase + ((size_t)id * KB + (kb0 + lf)) * 64);
                    n++;
                }
            if (++lf == nkb) { lf = 0; lb++; }
        }
        return n;
    };
#pragma unroll
    for (int s = 0; s < DEPTH; s++) cnt[s] = 0;
#pragma unroll
    for (int s = 0; s < DEPTH - 1; s++) cnt[s] = request(s);
    tl_stamp(a.tl, 1);
    {   // this wave's share of the image is older than its weights
        int newer = 0;
#pragma unroll
        for (int s = 0; s < DEPTH - 1; s++) newer += cnt[s];
        wait_vm_dyn<(DEPTH - 1) * 2 * NTW>(newer);
    }
    if (threadIdx.x < NH * SEQ_T) { recl[2 * threadIdx.x] = scale_of(rc.amax); recl[2 * threadIdx.x + 1] = rc.cA; }
    __syncthreads();
    tl_stamp(a.tl, 2);
    zero_acc();
    int cb = 0, cf = 0;                                           // compute cursor: batch, k-block
    for (int g0 = 0; g0 < total; g0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) {
            if (g0 + s >= total) break;                           // (wave-uniform)
            cnt[(s + DEPTH - 1) % DEPTH] = request((s + DEPTH - 1) % DEPTH);
            {
                int newer = 0;
#pragma unroll
                for (int t = 0; t < DEPTH; t++) if (t != s) newer += cnt[t];
                wait_vm_dyn<(DEPTH - 1) * 2 * NTW>(newer);
            }
            // load_b_asm / load_u32_asm hand out their destinations before the data is there: the registers CHANGE here
#pragma unroll
            for (int i = 0; i < NTW; i++) asm volatile("" : "+v"(bwr[s][i]));
            const int m = batch_mask(cb);
            if (cf == 0) {
#pragma unroll
                for (int i = 0; i < NTW; i++) { asm volatile("" : "+v"(rsn[i])); rsv[i] = rsn[i]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 *ab = abuf + lane + (size_t)cf * 384;
#pragma unroll
            for (int hh = 0; hh < NH; hh++) {
                u32x4 av[2][3];
#pragma unroll
                for (int ms = 0; ms < 2; ms++)
#pragma unroll
                    for (int b = 0; b < 3; b++) av[ms][b] = ab[(size_t)hh * nkbm * 384 + (ms * 3 + b) * 64];
#pragma unroll
                for (int i = 0; i < NTW; i++) {
                    if (!(m >> i & 1)) continue;
                    const u32x4 w = bwr[s][i];
                    const i32x4 bf = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
#pragma unroll
                    for (int ms = 0; ms < 2; ms++)
#pragma unroll
                        for (int b = 0; b < 3; b++) {
                            const i32x4 af = i32x4{(int)av[ms][b][0], (int)av[ms][b][1], (int)av[ms][b][2], (int)av[ms][b][3]};
                            acc[hh][i][ms][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[hh][i][ms][b], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++cf == nkb) {                                    // the batch's last block: its partial values out (the next batch's blocks are in flight)
                if (cb == 0) tl_stamp(a.tl, 6);
#pragma unroll
                for (int hh = 0; hh < NH; hh++)
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int i = 0; i < NTW; i++) {
                            if (!(m >> i & 1)) continue;
                            const int id = id0 + cb * bs + i;
                            const double *rl = recl + 2 * ((size_t)hh * SEQ_T + mt * 16 + 4 * (lane >> 4));
                            float *dst = a.pk + hh * a.pk_h + pk_lane_base(ntiles, j, id, mt, lane);
                            const size_t rst = pk_rstride(ntiles);
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const double M = (double)acc[hh][i][mt][0][r] + 256.0 * (double)acc[hh][i][mt][1][r] + 65536.0 * (double)acc[hh][i][mt][2][r];
                                dst[r * rst] = (float)(rl[2 * r] * (M + rl[2 * r + 1] + SEQ_CU * (double)rsv[i]));   // scale_o (M + cA_o + CU rowsum_o)
                            }
                        }
                zero_acc();
                cf = 0; cb++;
                if (cb == 1) tl_stamp(a.tl, 3);
            }
        }
    }
    tl_stamp(a.tl, 4);
    tl_stamp(a.tl, 5);
}

// ------------------------------------------------------------------------------------------
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
// K loop of one pass over NTL tiles: wave w multiplies the k-blocks [kb0, kb1) of ITS octant; operands of the next
// k-block(s) are requested before the MFMAs of the current one issue.
template <int NTL, int DEPTH>
__device__ __forceinline__ void seq_pass(i32x4 (&acc)[SEQ_TB][2][3], const u32x4 *const (&wt)[SEQ_TB], const u32x4 *img, int kb0, int kb1, int lane)
{
    const int n = kb1 - kb0;
    if (n <= 0) return;
    auto kbi = [&](int it) { return kb0 + (it < n - 1 ? it : n - 1); };   // past the end: re-read the last (branch-free loop)
    SeqFrag<NTL> f[DEPTH];
#pragma unroll
    for (int jj = 0; jj < DEPTH - 1; jj++) seq_frag_load<NTL>(f[jj], wt, img, kbi(jj), lane);
    int it = 0;
    for (; it + DEPTH <= n; it += DEPTH) {
#pragma unroll
        for (int jj = 0; jj < DEPTH; jj++) {
            seq_frag_load<NTL>(f[(jj + DEPTH - 1) % DEPTH], wt, img, kbi(it + jj + DEPTH - 1), lane);
            seq_frag_mfma<NTL>(f[jj], acc);
        }
    }
#pragma unroll
    for (int jj = 0; jj < DEPTH - 1; jj++)
        if (it + jj < n) seq_frag_mfma<NTL>(f[jj], acc);
}

// "K over the waves" (head): one workgroup owns a contiguous range of 16-row tiles and works through it in passes of up to
// SEQ_TB tiles; wave w takes octant w of K; the per-octant values (exact integers scaled in f64) meet in LDS, then the
// workgroup adds the offset term and stores.
__global__ __launch_bounds__(SEQ_NT) void k_seq_gemm_ks(SeqGemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float (*accl)[SEQ_TB][2][4][64] = reinterpret_cast<float (*)[SEQ_TB][2][4][64]>(smem);   // [octant][tile][row tile][reg][lane], 80 KiB
    float *sol = reinterpret_cast<float *>(smem + sizeof(float) * SEQ_O * SEQ_TB * 2 * 4 * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, KB = K >> 6, N = a.N, Q = a.Q;
    const int nch = (N + Q - 1) / Q;                    // channels per class
    const int CB = (nch + 15) >> 4;                     // 16-channel blocks per class
    const int ntiles = Q * CB;
    const int tb0 = block_lo(ntiles), tb1 = block_hi(ntiles);
    const int kb0 = (int)(((long long)wave * KB) / SEQ_O), kb1 = (int)(((long long)(wave + 1) * KB) / SEQ_O);
    if (blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < a.cp_n; j += SEQ_NT) a.cp_dst[j] = a.cp_src[j];

    int tg = tb0;
    int vcur = -1;
    while (tg < tb1) {
        // tiles of this pass: same activation vector
        const int v0 = a.vec_of_q[tg / CB];
        int nt = 1;
        while (nt < SEQ_TB && tg + nt < tb1 && a.vec_of_q[(tg + nt) / CB] == v0) nt++;
        if (v0 != vcur && threadIdx.x < a.T) sol[threadIdx.x] = seq_so(a.part, v0, threadIdx.x);
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
        case 1: seq_pass<1, 4>(acc, wt, img, kb0, kb1, lane); break;
        case 2: seq_pass<2, 3>(acc, wt, img, kb0, kb1, lane); break;
        case 3: seq_pass<3, 3>(acc, wt, img, kb0, kb1, lane); break;
        case 4: seq_pass<4, 2>(acc, wt, img, kb0, kb1, lane); break;
        default: seq_pass<5, 2>(acc, wt, img, kb0, kb1, lane); break;
        }
        // this wave's octant: fold the limbs (exact in f64), scale with the octant's records; the octants meet in LDS as f32
        // values and are added in a fixed order below (deterministic; the same arithmetic as k_seq_gemm_p's per-slice partials)
#pragma unroll
        for (int i = 0; i < SEQ_TB; i++)
            if (i < nt) {
                const int id = tg + i, q = id / CB, ch = 16 * (id % CB) + (lane & 15), row = Q * ch + q;
                const bool rok = ch < nch && row < N;
                const unsigned rs = rok ? a.rs8[(size_t)wave * N + row] : 0u;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int t = mt * 16 + 4 * (lane >> 4) + r;
                        float val = 0.f;
                        if (t < a.T && rok) {
                            const SeqPart rc = a.part[((size_t)v0 * SEQ_T + t) * SEQ_O + wave];
                            const double M = (double)acc[i][mt][0][r] + 256.0 * (double)acc[i][mt][1][r] + 65536.0 * (double)acc[i][mt][2][r];
                            val = (float)seq_slice_value(rc, M, rs);
                        }
                        accl[wave][i][mt][r][lane] = val;
                    }
            }
        __syncthreads();
        // epilogue: D[m][n] with n = lane & 15 (weight row of the tile), m = 4 * (lane >> 4) + reg (row of the chunk)
        for (int e = threadIdx.x; e < nt * 2 * 4 * 64; e += SEQ_NT) {
            const int ln = e & 63, r = (e >> 6) & 3, mt = (e >> 8) & 1, i = e >> 9;
            const int t = mt * 16 + 4 * (ln >> 4) + r;
            const int id = tg + i, q = id / CB, ch = 16 * (id % CB) + (ln & 15);
            const int row = Q * ch + q;
            if (t < a.T && ch < nch && row < N) {
                double v = 0.0;
#pragma unroll
                for (int o = 0; o < SEQ_O; o++) v += (double)accl[o][i][mt][r][ln];
                a.out[(size_t)t * N + row] = (float)v + sol[t];
            }
        }
        __syncthreads();
        tg += nt;
    }
}


} // namespace rwkvk
