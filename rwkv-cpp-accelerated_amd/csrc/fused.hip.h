// fused.hip.h -- two decode kernels of a layer as ONE launch with an in-kernel hand-off: k_attout -> k_ffn_rk.
//
// Why this pair, and why it can pay where the one-launch token did not.  A launch of this engine costs ~5 us besides
// its stream (start-up 1.0, landing + finishing the last group 2, spread between workgroups 0.9, boundary 1.3), and k_attout moves
// only 64 KiB per CU: 6.9 us for 2.6 us of streaming.  An in-kernel all-to-all costs MORE than a kernel boundary (7+ us: drain,
// arrival, poll, ~80 KB of agent-scope re-reads per CU, reduce, quantise, stage) -- unless the memory system does useful work
// meanwhile.  Weights do not depend on activations: while the chip hands x over, the loader wave keeps streaming k_ffn_rk's rows
// (320 KiB per CU) into the LDS ring, and the consumer waves that do not run the prologue copy their first phase-2 group out of the
// ring into registers, so the ring turns over: ~180 KiB of run-ahead, ~7 us of stream, covers the seam.  The one-launch token had
// the ring alone (100 KiB) against 10-14 us edges.  Arithmetic, summation order and every output are those of the two separate
// kernels: results are bit-identical, the tests of the launch kernels apply unchanged.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, R1): phase-1 outputs that phase 2 reads on OTHER CUs -- x, B_m, the
// workgroup tuples -- are stored write-through (agent scope); every consumer wave drains its stores before it counts itself in;
// the workgroup's wave 0 publishes the tuple, drains, and adds 1 to the arrival counter of its shard (blockIdx & 7; 64 B apart).
// Wave 0 of every workgroup polls the 8 shards (lanes 0..7, relaxed agent-scope loads, s_sleep), then the prologue waves read
// tuples / x / B with agent-scope loads (the producers stored write-through: no cache invalidate).  The counters are zeroed by the
// NEXT layer's launch of this kernel (its predecessor has drained by then: kernel boundary), the first one by the last.
// Every wait is bounded; all 256 workgroups are co-resident by construction (one per CU, launched on a drained stream).
#pragma once
#include "kernels.hip.h"

namespace rwkvk {

struct EdgeSync {
    unsigned *cnt;        // this launch's arrival counters: shard s at cnt[16 * s]
    unsigned *cnt_zero;   // the counters this launch zeroes (the previous layer's)
};
constexpr int EDGE_SPIN = 1 << 20;

// 16-byte agent-scope (sc1) load: bypasses the L1, which another CU's stores never refresh
__device__ __forceinline__ u32x4 ld_sc16(const void *base, unsigned byte_off)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
}
template <class T> __device__ __forceinline__ void st_sc(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// consumer waves (wave < NC) of a two-phase kernel, behind their last phase-1 group: fold the site accumulators into the
// workgroup's tuple (site_publish's arithmetic over the NC consumer waves; the loader wave holds no rows), publish it, arrive.
// scratch: NC * LIVE * 12 words of LDS nobody else uses.
template <int NV, int LIVE>
__device__ __forceinline__ void edge_publish(SiteAcc<NV> &acc, const SiteDyn &dy, void *scratch, GldsCtl *gc, const EdgeSync &es, unsigned &fail)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NE = NC * LIVE;
    double *red = reinterpret_cast<double *>(scratch);
    float *redf = reinterpret_cast<float *>(red + NE * 8);
    if (lane < LIVE) {
        const int e = w * LIVE + lane;
#pragma unroll
        for (int k = 0; k < 8; k++) red[e * 8 + k] = acc.d[k];
#pragma unroll
        for (int k = 0; k < 4; k++) redf[e * 4 + k] = acc.f[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores (x, B, state) have been acknowledged
    if (lane == 0) __hip_atomic_fetch_add(&gc->pub, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (w != 0) return;
    wait_count(&gc->pub, NC, fail);
    if (lane < 8) {
        double t = 0.0;
        for (int i = 0; i < NE; i++) t += red[i * 8 + lane];
        st_sc(dy.pd + (size_t)blockIdx.x * 8 + lane, t);
    } else if (lane < 12) {
        float t = 0.f;
        for (int i = 0; i < NE; i++) t = fmaxf(t, redf[i * 4 + (lane - 8)]);
        st_sc(dy.pf + (size_t)blockIdx.x * 4 + (lane - 8), t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&es.cnt[16 * (blockIdx.x & 7u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: every workgroup of the grid has arrived (lanes 0..7 watch one shard each)
__device__ __forceinline__ void edge_poll(const EdgeSync &es, int lane, unsigned &fail)
{
    const unsigned s = (unsigned)lane & 7u, want = (gridDim.x + 7u - s) >> 3;
    bool done = false;
    for (int it = 0; it < EDGE_SPIN; it++) {
        unsigned v = want;
        if (lane < 8) v = __hip_atomic_load(&es.cnt[16 * s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64((int)(v - want) < 0) == 0ull) { done = true; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    fail = done ? fail : 5u;
    asm volatile("" ::: "memory");
}

// LayerNorm-site prologue of PHASE 2 (prologue waves 0..NWP-1; ring_site's arithmetic): static inputs are requested first, then the
// chip-wide hand-off is awaited (wave 0 polls, the others watch an LDS word), then tuples / x / B are read with agent-scope loads.
template <int NV, int S>
__device__ __forceinline__ void edge_site(const SiteStatic &st, const SiteDyn &dy, const double *x, int D, double *red, unsigned *xq, bool publish_stats,
                                          GldsCtl *gc, const EdgeSync &es, unsigned long long *tl, unsigned &fail)
{
    constexpr int NTP = NT / 2, NWP = NTP / 64, NQP = (S * 256 + NTP - 1) / NTP;
    const int nqd = D >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    double tc[NV];
    float mc[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { tc[m] = st.TC[m]; mc[m] = st.maxC[m]; }
    f32x4 Cq[NQP][NV];
#pragma unroll
    for (int i = 0; i < NQP; i++) {
        const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
#pragma unroll
        for (int m = 0; m < NV; m++) Cq[i][m] = reinterpret_cast<const f32x4 *>(st.C + (size_t)m * D)[qc];
    }
    if (wave == 0) {
        edge_poll(es, lane, fail);
        if (lane == 0) __hip_atomic_store(&gc->go, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        wait_count(&gc->go, 1u, fail);
    }
    tl_stamp(tl, 3);
    SiteTuple tup;
    {
        const unsigned i = (int)threadIdx.x < dy.n_part ? threadIdx.x : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) tup.d[k] = __builtin_bit_cast(f64x2, ld_sc16(dy.pd, i * 64u + k * 16u));
        tup.f = __builtin_bit_cast(f32x4, ld_sc16(dy.pf, i * 16u));
    }
    double xl[NQP][4];
    f32x4 Bq[NQP][NV];
#pragma unroll
    for (int i = 0; i < NQP; i++) {
        const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
        const f64x2 xa = __builtin_bit_cast(f64x2, ld_sc16(x, (unsigned)qc * 32u)), xb = __builtin_bit_cast(f64x2, ld_sc16(x, (unsigned)qc * 32u + 16u));
        xl[i][0] = xa[0]; xl[i][1] = xa[1]; xl[i][2] = xb[0]; xl[i][3] = xb[1];
#pragma unroll
        for (int m = 0; m < NV; m++) Bq[i][m] = __builtin_bit_cast(f32x4, ld_sc16(dy.B, ((unsigned)m * (unsigned)D + (unsigned)qc * 4u) * 4u));
    }
    SiteRed<NV> r;
    site_reduce<NV, NWP>(st, dy, tup, D, red, r, tc, mc, nullptr, &gc->spin2);
    if (publish_stats && blockIdx.x == 0 && threadIdx.x == 0) { dy.lnstat[0] = r.mean; dy.lnstat[1] = r.rstd; }
    tl_stamp(tl, 4);
    site_stage<NV, NQP, S, NTP>(xl, Cq, Bq, r, xq, nqd);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int m = 0; m < NV; m++) { bc[m] = (float)r.S[m]; bc[4 + m] = r.amax[m]; }
    }
    if (lane == 0) __hip_atomic_fetch_add(&gc->staged2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ------------------------------------------------------------------------------------------
// k_attout (kernels.hip.h) and k_ffn_rk as the two phases of one launch.  LDS: reduction scratch | staged vector of phase 1 |
// the two staged vectors of phase 2 | publish scratch | ring control | ring (units of one row of S KiB, shared by both phases).
constexpr int FUSE_PSCR = 2048;
template <int S> __host__ __device__ constexpr size_t fuse_fixed() { return (size_t)RED_BYTES + 3 * (size_t)S * 3072 + FUSE_PSCR; }
template <int S, int R>
__global__ __launch_bounds__(NT) void k_attout_ffn_rk(AttOutArgs a, FfnRKArgs f, EdgeSync es)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq1 = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    unsigned *xq2 = reinterpret_cast<unsigned *>(smem + RED_BYTES + (size_t)S * 3072);
    unsigned char *pscr = smem + RED_BYTES + 3 * (size_t)S * 3072;
    GldsCtl *gc = reinterpret_cast<GldsCtl *>(smem + fuse_fixed<S>());
    unsigned char *ring = reinterpret_cast<unsigned char *>(gc + 1);
    const int D = a.D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = D >> 4;
    tl_stamp(f.tl, 0);
    if (blockIdx.x == 0 && threadIdx.x < 8) es.cnt_zero[16 * threadIdx.x] = 0u;     // visible to its next user behind this kernel's end
    // phase 1: groups of R att_out rows; phase 2: channels (4 ffn_k rows + 1 ffn_r row)
    const int G1 = (D + R - 1) / R;
    const int g10 = (int)(((long long)blockIdx.x * G1) / gridDim.x), g11 = (int)(((long long)(blockIdx.x + 1) * G1) / gridDim.x);
    const int g20 = (int)(((long long)blockIdx.x * D) / gridDim.x), g21 = (int)(((long long)(blockIdx.x + 1) * D) / gridDim.x);
    auto base1 = [&](int gg) {
        int row = (gg < g11 ? gg : (g11 > g10 ? g11 - 1 : 0)) * R;
        if (row > D - R) row = D - R;
        return a.w + (size_t)row * D;
    };
    auto base2 = [&](int g) { return f.w + (size_t)(g < g21 ? g : (g21 > g20 ? g21 - 1 : 0)) * 5 * D; };
    unsigned fail = 0u;
    double part = 0.0;
    float pmax = 0.f;
    if (wave == NC) {
        // ---- the loader: both phases' rows back to back; it never looks at the seam ----
        for (int i = lane; i < (int)(sizeof(GldsCtl) / 4); i += 64) reinterpret_cast<unsigned *>(gc)[i] = 0u;
        RingLoader<S> ld(gc, lds_addr(ring), a.ns, chunks, lane);
        int g = g10;
        const int pre = RWKV_RING_PRE < a.ns - R ? RWKV_RING_PRE : a.ns - R;
        for (; g < g11 && (int)ld.issued < pre; g++) ld.template group<R>(base1(g) + ld.off[0], (size_t)D, true);
        __syncthreads();   // order: the control block is zero, the phase-1 prologue's requests are in the pipe
        for (; g < g11; g++) ld.template group<R>(base1(g) + ld.off[0], (size_t)D);
#ifdef RWKV_FUSE_THIN      // experiment: a thin stream (RWKV_RING_PRE_DEPTH pieces in flight) until the phase-2 vectors are staged
        for (g = g20; g < g21; g++) {
            const bool thin = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&gc->staged2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < NT / 128;
            ld.template group<5>(base2(g) + ld.off[0], (size_t)D, thin);
        }
#else
        for (g = g20; g < g21; g++) ld.template group<5>(base2(g) + ld.off[0], (size_t)D);
#endif
        ld.finish();
        fail = ld.fail;
        tl_stamp(f.tl, 2);
    } else {
        // ---- phase 1: att_out (k_attout's pre / epi with write-through stores of what phase 2 reads on other CUs) ----
        const double mean1 = a.lnstat[0], rstd1 = a.lnstat[1];
        const size_t so = (size_t)a.ctl->slot * a.slot_stride;
        float Sf, amax;
        ring_vec<1, S>(a.ybuf, a.partS, a.partM, a.n_part, D, red, xq1, Sf, amax, gc, nullptr, fail);
        const double sc = scale_of(amax);
        SiteAcc<2> acc;
        acc.clear();
        auto pre1 = [&](int g) {
            AttOutIn<R> in;
            int row0 = g * R;
            in.shift = (row0 > D - R) ? row0 - (D - R) : 0;
            row0 -= in.shift;
            in.mi = row0 + (lane < R ? lane : 0);
            in.rsum = a.rs[in.mi];
            in.xold = a.x[in.mi]; in.lw = a.lnw[in.mi]; in.lb = a.lnb[in.mi]; in.prev2 = a.sdd[so + in.mi];
            site_prefetch<2>(a.st, in.mi, in.pre);
            return in;
        };
        auto epi1 = [&](int, const unsigned long long (&T)[R], const AttOutIn<R> &in) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (lane == r && r >= in.shift) {
                    const float accf = (float)in.xold + (row_value(T[r], in.rsum, sc) + Sf);
                    const double xnew = (double)accf;
                    st_sc(a.x + in.mi, xnew);
                    a.sxy[so + in.mi] = in.lw * ((in.xold - mean1) * rstd1) + in.lb;
                    site_emit<2, true>(in.pre, a.dy, D, in.mi, xnew, in.prev2, acc);
                }
            }
        };
        ring_groups<R, S, PAT_SHARED>(g10, g11, a.ns, ring, gc, xq1, lane, wave, chunks, pre1, epi1, fail);
        tl_stamp(f.tl, 1);
        edge_publish<2, R>(acc, a.dy, pscr, gc, es, fail);
        tl_stamp(f.tl, 6);
        // ---- phase 2: ffn r / k (k_ffn_rk's pre / epi) ----
        double sck = 0.0, scr = 0.0;
        float Sk = 0.f, Sr = 0.f;
        auto pre2 = [&](int g) {
            FfnRKIn in;
            in.rsum = f.rs[g * 5 + (lane < 5 ? lane : 0)];
            const int kk = 4 * g + (lane < 4 ? lane : 0);
            in.rq = f.r_fv[kk]; in.oq = f.o_fv[kk];
            return in;
        };
        auto epi2 = [&](int g, const unsigned long long (&T)[5], const FfnRKIn &in) {
            float val = 0.f;
#pragma unroll
            for (int r = 0; r < 5; r++) {
                const float vr = row_value(T[r], in.rsum, r < 4 ? sck : scr) + (r < 4 ? Sk : Sr);
                val = lane == r ? vr : val;
            }
            if (lane < 4) {
                float h = val * (float)(val > 0.f);
                h = h * h;
                const float hs = h * in.rq;
                f.hbuf[4 * g + lane] = hs;
                part += (double)(h * in.oq);
                pmax = fmaxf(pmax, fabsf(hs));
            } else if (lane == 4) {
                f.rgate[g] = (float)(1.0 / (1.0 + exp(-(double)val)));
            }
        };
        // the phase-2 vectors are staged by waves 0..3 once the chip has handed x over; everybody else meets them at `staged2`
        auto gate = [&]() {
            const float *bc = reinterpret_cast<const float *>(red + RED_BC);
            wait_count(&gc->staged2, NT / 128, fail);
            sck = scale_of(bc[4]); scr = scale_of(bc[5]); Sk = bc[0]; Sr = bc[1];
            tl_stamp(f.tl, 5);
        };
        const unsigned ub = (unsigned)(g11 - g10) * R, kb = (unsigned)(g11 - g10);
        if (wave < NT / 128) {
            edge_site<2, S>(f.st, f.dy, f.x, D, red, xq2, true, gc, es, f.tl, fail);
            gate();
            ring_groups<5, S, PAT_FFN_RK>(g20, g21, a.ns, ring, gc, xq2, lane, wave, chunks, pre2, epi2, fail, nullptr, ub, kb);
        } else {
            ring_groups<5, S, PAT_FFN_RK>(g20, g21, a.ns, ring, gc, xq2, lane, wave, chunks, pre2, epi2, fail, nullptr, ub, kb, gate);
        }
    }
    __syncthreads();   // every wave is past its last read of the reduction scratch
    block_sum_max(part, pmax, red + RED_PART);
    if (threadIdx.x == 0) { f.partS[blockIdx.x] = part; f.partM[blockIdx.x] = pmax; }
    ring_report(fail, f.herr);
    tl_stamp(f.tl, 7);
}

} // namespace rwkvk
