// sampler.hip.h -- typical sampling over the 50277 logits ON THE DEVICE (gfx950), so that the reference's
// generation loop (examples/storygen/storygen.cpp:63-69: out[0] = -99; typical(out, temp, tau)) needs no
// 201 KB logits download and no host sort per token.
//
// Behavioural mirror of reference include/rwkv/sampler/typical.h:20-58.  Two modes (include/rwkv_sampler.h has the
// findings): recipe = 0 (default) is what the reference COMPUTES -- a draw from softmax^n, n = uint8(1/temp), its cut at
// tau being a no-op -- and needs no selection at all; recipe = 1 is the recipe its header comment documents (softmax,
// entropy H, sort by |-log p - H|, keep the smallest prefix whose cumulative probability reaches tau, p^(1/temp), draw).
// For the latter the sort is not needed: with s_i = |-log p_i - H| the cut-off value is
//     thr = min { v in {s_i} : sum_{s_i <= v} p_i >= tau }
// (the sorted prefix crosses tau exactly at the first element of that value), found as a 32-bit float
// pattern by a three-pass radix select over probability-mass histograms in LDS.  Three launches:
// k_typical_stats (64 workgroups: online-softmax partials), k_typical_keys (64 workgroups: p_i and s_i as
// f32, every sum in f64) and k_typical (one workgroup of 1024 threads: select, weights, draw).  The draw is the inverse CDF in token order for a uniform
// u supplied by the caller or derived from (seed, step) with splitmix64 -- include/rwkv_sampler.h has the
// same deterministic draw on the host (typical_u), which is what the parity test compares against.
#pragma once
#include "kernels.hip.h"

namespace rwkvk {

constexpr int TS_NT = 1024;                // threads of the selection kernel
constexpr int TS_PER = 50;                 // tokens per thread there: 1024 * 50 = 51200 >= 50277
constexpr int TS_HIST = 4096;              // mass-histogram bins of a radix-select pass (12 bits)
constexpr int TS_G = 64;                   // workgroups of the two element-wise kernels
constexpr int TS_GT = 256;                 // threads of those
static_assert(TS_NT * TS_PER >= (int)VOCAB, "sampler tiling must cover the vocabulary");

struct TypicalArgs {
    const float *logits;          // [rows][V]
    int row;                      // logits row; < 0: ctl->out_row
    Ctl *ctl;                     // token / step feedback (device-side generation loop)
    unsigned long long *gen;      // generated ids, gen[step]
    unsigned gen_cap;
    float temp, tau;
    int recipe;                   // 1: the recipe typical.h documents (cut at tau, p^(1/temp)); 0: what it computes (no cut, p^uint8(1/temp))
    double expo;                  // the exponent applied to p
    double u;                     // uniform in [0, 1) when use_seed == 0
    unsigned long long seed;      // else u = uniform(splitmix64(seed + step))
    int use_seed;
    int ban0;                     // logits[0] = -99 before sampling (storygen.cpp:66)
    int feedback;                 // write the pick into ctl->token and advance ctl->step
    unsigned long long *pick;     // the sampled id (host-visible copy source)
    double *part;                 // [TS_G][3] per-workgroup (max, sum exp(l - max), sum exp(l - max) (l - max))
    float *p;                     // [TS_NT * TS_PER] probabilities, position (i % TS_PER) * TS_NT + i / TS_PER for token i
    unsigned *key;                // same layout: bit patterns of |-log p - H|
    float *pw;                    // same layout: p^(1/temp) (typical.h:49-52), zeroed for the tokens that are cut
};

__device__ __forceinline__ const float *ts_row(const TypicalArgs &a) { return a.logits + (size_t)(a.row >= 0 ? (unsigned)a.row : a.ctl->out_row) * VOCAB; }
__device__ __forceinline__ float ts_logit(const TypicalArgs &a, const float *lg, int i) { return (a.ban0 && i == 0) ? -99.0f : lg[i]; }

// (1) per-workgroup online-softmax partials over a slice of the vocabulary
__global__ __launch_bounds__(TS_GT) void k_typical_stats(TypicalArgs a)
{
    __shared__ double red[RED_BYTES / 8];
    const int V = (int)VOCAB;
    const int i0 = block_lo(V), i1 = block_hi(V);
    const float *lg = ts_row(a);
    float mx[1] = {-INFINITY};
    for (int i = i0 + threadIdx.x; i < i1; i += TS_GT) mx[0] = fmaxf(mx[0], ts_logit(a, lg, i));
    {   // workgroup max (TS_GT / 64 waves)
        float *rf = reinterpret_cast<float *>(red);
        const float w = wave_max(mx[0]);
        if ((threadIdx.x & 63) == 0) rf[threadIdx.x >> 6] = w;
        __syncthreads();
        float m = -INFINITY;
        for (int k = 0; k < TS_GT / 64; k++) m = fmaxf(m, rf[k]);
        mx[0] = m;
        __syncthreads();
    }
    double z = 0.0, sl = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += TS_GT) {
        const double d = (double)ts_logit(a, lg, i) - (double)mx[0];
        const double e = exp(d);
        z += e; sl += e * d;
    }
    z = wave_sum(z); sl = wave_sum(sl);
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = z; red[(threadIdx.x >> 6) * 2 + 1] = sl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tz = 0.0, ts = 0.0;
        for (int k = 0; k < TS_GT / 64; k++) { tz += red[2 * k]; ts += red[2 * k + 1]; }
        a.part[blockIdx.x * 3 + 0] = (double)mx[0]; a.part[blockIdx.x * 3 + 1] = tz; a.part[blockIdx.x * 3 + 2] = ts;
    }
}

// combine the partials: M = max, Z = sum exp(l - M), H = entropy = log Z - sum p (l - M)   (typical.h:29-31)
__device__ __forceinline__ void ts_combine(const TypicalArgs &a, double &M, double &logZ, double &H)
{
    const int lane = threadIdx.x & 63;
    const double m = lane < TS_G ? a.part[lane * 3] : -INFINITY, z = lane < TS_G ? a.part[lane * 3 + 1] : 0.0, sl = lane < TS_G ? a.part[lane * 3 + 2] : 0.0;
    double mm = m;
    for (int off = 32; off >= 1; off >>= 1) mm = fmax(mm, __shfl_xor(mm, off, 64));
    const double sc = lane < TS_G ? exp(m - mm) : 0.0;
    const double Z = wave_sum(sc * z), S = wave_sum(sc * (sl + (m - mm) * z));
    M = mm; logZ = log(Z); H = logZ - S / Z;
}
static_assert(TS_G <= 64, "one lane per partial");

// (2) probabilities and keys, written in the selection kernel's thread-contiguous order
__global__ __launch_bounds__(TS_GT) void k_typical_keys(TypicalArgs a)
{
    const int V = (int)VOCAB;
    const int i0 = block_lo(V), i1 = block_hi(V);
    const float *lg = ts_row(a);
    double M, logZ, H;
    ts_combine(a, M, logZ, H);
    const double it = a.expo;
    for (int i = i0 + threadIdx.x; i < i1; i += TS_GT) {
        const double nl = (M - (double)ts_logit(a, lg, i)) + logZ;          // -log p_i
        const int pos = (i % TS_PER) * TS_NT + i / TS_PER;
        a.p[pos] = (float)exp(-nl);
        a.pw[pos] = it == 1.0 ? (float)exp(-nl) : it == 0.0 ? 1.0f : (float)exp(-nl * it);   // p^expo (p^0 = 1 even for p = 0, as nc::power), here: 64 workgroups share the exponentials
        a.key[pos] = __float_as_uint((float)fabs(nl - H));                  // typical.h:32
    }
    if (blockIdx.x == 0)       // padding positions: tokens V .. TS_NT * TS_PER - 1 carry no mass and the largest key
        for (int i = V + threadIdx.x; i < TS_NT * TS_PER; i += TS_GT) {
            const int pos = (i % TS_PER) * TS_NT + i / TS_PER;
            a.p[pos] = 0.f; a.pw[pos] = 0.f; a.key[pos] = 0x7f800000u;
        }
}

__device__ __forceinline__ double ts_block_sum(double v, double *red)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v = wave_sum(v);
    __syncthreads();              // previous users of `red` are done
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < TS_NT / 64; i++) s += red[i];
    return s;
}

// (3) threshold by radix select, weights, inverse-CDF draw.  Thread t owns tokens [50 t, 50 t + 50); its k-th
// token sits at position k * TS_NT + t, so every pass over the (L2-resident) arrays is coalesced.
__global__ __launch_bounds__(TS_NT) void k_typical(TypicalArgs a)
{
    __shared__ double red[TS_NT / 64];
    __shared__ double scan[TS_NT];
    __shared__ double hist[TS_HIST];
    __shared__ int sel_bin;
    __shared__ double sel_before;
    __shared__ unsigned pick_s;
    const int t = threadIdx.x, i0 = t * TS_PER;

    // thr = min { v : sum_{key <= v} p >= tau }, found as a bit pattern (monotone for non-negative floats) by a
    // most-significant-digit-first radix select over mass histograms in LDS: 12 + 12 + 8 bits.  Tokens whose
    // probability cannot move an f64 sum (< 1e-30) are skipped.  If the total never reaches tau the threshold
    // is +inf: everything is kept (typical.h: the cutoff clamps to the last index).
    const double tau = fmax((double)a.tau, 1e-300);   // tau <= 0 keeps the smallest key, as the reference's cutoff = 0 does
    unsigned prefix = 0u;
    double acc = 0.0;                                  // mass of all keys below the current prefix range
    bool open_end = !a.recipe;                         // as compiled the reference cuts nothing (typical.h:50): no selection
#pragma unroll 1
    for (int pass = 0; pass < (a.recipe ? 3 : 0); pass++) {
        const int shift = pass == 0 ? 20 : pass == 1 ? 8 : 0, bits = pass == 2 ? 8 : 12, nb = 1 << bits;
        for (int b = t; b < TS_HIST; b += TS_NT) hist[b] = 0.0;
        __syncthreads();
        // unrolled by 10: twenty loads in flight per thread (one load per trip would expose an L2 latency per token)
#pragma unroll 10
        for (int k = 0; k < TS_PER; k++) {
            const unsigned kb = a.key[k * TS_NT + t];
            const float pk = a.p[k * TS_NT + t];
            const bool in = pass == 0 || (kb >> (shift + bits)) == (prefix >> (shift + bits));
            if (in && pk > 1e-30f) __hip_atomic_fetch_add(&hist[(kb >> shift) & (nb - 1)], (double)pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        // wave 0: lane l owns bins [l * per, (l + 1) * per); find the first bin where the running mass reaches tau
        if (t < 64) {
            const int per = nb / 64;
            double loc = 0.0;
            for (int b = 0; b < per; b++) loc += hist[t * per + b];
            double incl = loc;                         // inclusive scan over the 64 lanes
            for (int off = 1; off < 64; off <<= 1) {
                const double v = __shfl_up(incl, off, 64);
                if (t >= off) incl += v;
            }
            double run = acc + (incl - loc);
            int found = 0x7fffffff;
            double bef = 0.0;
            for (int b = 0; b < per; b++) {
                const double h = hist[t * per + b];
                if (found == 0x7fffffff && h > 0.0 && run + h >= tau) { found = t * per + b; bef = run; }
                run += h;
            }
            int best = found;
            for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
            if (found == best && best != 0x7fffffff) { sel_bin = best; sel_before = bef; }
            if (t == 0 && best == 0x7fffffff) sel_bin = -1;
        }
        __syncthreads();
        if (sel_bin < 0) { open_end = true; break; }
        prefix |= (unsigned)sel_bin << shift;
        acc = sel_before;
        __syncthreads();
    }
    const unsigned thr = open_end ? 0x7f800000u : prefix;
    // weights of the kept set (already raised to 1/temp) and the inverse-CDF draw in token order
    double wsum = 0.0;
#pragma unroll 10
    for (int k = 0; k < TS_PER; k++) {
        const float w = a.key[k * TS_NT + t] <= thr ? a.pw[k * TS_NT + t] : 0.f;
        a.pw[k * TS_NT + t] = w;                       // own positions only
        wsum += (double)w;
    }
    scan[t] = wsum;
    __syncthreads();
    for (int off = 1; off < TS_NT; off <<= 1) {        // inclusive scan of the per-thread totals: Hillis-Steele in LDS
        const double v = t >= off ? scan[t - off] : 0.0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const double total = scan[TS_NT - 1], before = scan[t] - wsum;
    double u = a.u;
    if (a.use_seed) {
        unsigned long long x = a.seed + (unsigned long long)a.ctl->step + 0x9E3779B97F4A7C15ull;   // splitmix64
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
    const double target = u * total;
    if (t == 0) pick_s = 0xffffffffu;
    __syncthreads();
    // the owner of the target: before <= target < before + wsum
    if (wsum > 0.0 && target >= before && target < before + wsum) {
        double c = before;
        int sel = -1, lastkept = -1;
#pragma unroll 10
        for (int k = 0; k < TS_PER; k++) {
            const float w = a.pw[k * TS_NT + t];
            if (w > 0.f) { c += (double)w; lastkept = i0 + k; if (sel < 0 && target < c) sel = i0 + k; }
        }
        if (sel < 0) sel = lastkept;               // rounding at the upper edge of this thread's range
        atomicMin(&pick_s, (unsigned)sel);
    }
    __syncthreads();
    if (t == 0) {
        unsigned sel = pick_s;
        if (sel == 0xffffffffu) {                  // target == total by rounding, or a degenerate distribution
            sel = 0;
        }
        if (a.pick) *a.pick = sel;
        if (a.feedback) {
            const unsigned st = a.ctl->step;
            if (st < a.gen_cap) a.gen[st] = sel;
            a.ctl->token = sel;
            a.ctl->step = st + 1;
        }
    }
}

} // namespace rwkvk
