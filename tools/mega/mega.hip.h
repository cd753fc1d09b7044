// mega.hip.h -- the whole token as ONE launch (k_token): the five decode kernels of kernels.hip.h as phases of a persistent
// 256-workgroup kernel, replacing the 4L + 2 dependent launches of rwkv.cu:528-589's per-token schedule.
//
// Why: a decode kernel streams its weights at the CU's load rate from its 2nd microsecond to its last row, and
// then ~4 us pass before the next kernel's stream is up again (tail, boundary, dispatch, cold arguments, first requests).
// With 4L + 2 such edges a 7B token spends a quarter of its time with HBM idle.  Here the stream never belongs to a phase:
//   * wave 7 of every workgroup is the LOADER.  It walks the token's phase list (first | att, att_out, ffn_rk, ffn_v per
//     layer | head) and DMAs this workgroup's weight rows of phase after phase into an LDS ring (global_load_lds_dwordx4,
//     1 KiB per instruction, <= 63 in flight), stopping only when the ring (~100 KiB of the CU's 160) is full.  Weights do
//     not depend on activations: while the chip finishes phase p and hands its vector over, the rings fill with phase p+1.
//   * waves 0..6 are CONSUMERS: group k of a phase (R rows = R ring units) belongs to wave k % 7; it waits for the loader's
//     `landed` count, copies the rows into registers, frees the units and runs group_dot + the row epilogue of kernels.hip.h.
//   * an EDGE (phase p -> p+1) replaces the kernel boundary.  Producer side: epilogues put their cross-workgroup outputs
//     (ybuf | x, B | hbuf) into LDS; the LAST consumer wave to finish adds up the waves' partial tuples and writes tuple and
//     outputs with agent-scope (sc1, write-through) stores, drains vmcnt and adds 1 to the arrival counter of its shard
//     (8 counters, 64 B apart: 32 arrivals each).  Consumer side: wave 0 polls the 8 counters (one relaxed sc1 load per
//     lane, s_sleep between polls), releases waves 1..3 through LDS, and the four prologue waves read tuples / x / B with
//     sc1 loads (the L1 is bypassed; no cache invalidate, no fence), reduce, quantise and stage exactly as the launch
//     kernels do.  Counters are monotonic (epoch * phases + phase), nothing is re-zeroed between tokens.
//   * every wait is bounded; a wait that gives up raises MegaSync::error (the host then fails the call and resets the block).
// All 256 workgroups must be co-resident: grid <= compute units, one workgroup per CU (the LDS footprint guarantees it).
// Numerics are those of the launch kernels (same staging, same integer dot products, same epilogues, same tuple order).
#pragma once
#include "../../rwkv-cpp-accelerated_amd/csrc/kernels.hip.h"

namespace rwkvk {

constexpr int MG_NC = NW - 1;          // consumer waves; wave NW-1 is the loader
constexpr int MG_NP = 4;               // prologue waves 0..3 (256 threads)
constexpr int MG_FQ = 32;              // groups the ring bookkeeping can hold
constexpr int MG_MAXROWS = 128;        // channels / rows of x a workgroup may own
constexpr int MG_SPIN = 1 << 17;       // bound of every wait loop
constexpr int MG_AR = 2;               // rows per group, att_out
constexpr int MG_HR = 4;               // rows per group, head
constexpr int MG_TL = 8;               // timeline stamps per (workgroup, phase)
// Loader depth policy.  A CU's vector-memory path returns data in order: every global access of an EDGE -- the drain of the
// hand-off stores, the arrival poll, the hand-off loads -- queues behind whatever DMA this CU's loader has in flight
// (63 KiB = 2.5 us).  So the loader streams deep only for the phase its consumers are in; when it runs AHEAD of them
// (prefetching phase p+1 while the workgroup finishes p and gathers) it keeps at most RWKV_MG_THINQ pieces in flight, and
// with RWKV_MG_PAUSE it issues nothing between "all consumer waves done" and "hand-off gathered".
#ifndef RWKV_MG_THINQ
#define RWKV_MG_THINQ 16
#endif
#ifndef RWKV_MG_PAUSE
#define RWKV_MG_PAUSE 1
#endif
#ifndef RWKV_MG_DEPTH
#define RWKV_MG_DEPTH 63               // pieces in flight when streaming for the consumers' own phase (<= 63: the counter has 6 bits)
#endif
#ifndef RWKV_MG_UNITPOLL
#define RWKV_MG_UNITPOLL 1             // read the DMA counter after every unit (0: only while waiting; landings are then announced with a fixed lag)
#endif
#ifndef RWKV_MG_NODOT
#define RWKV_MG_NODOT 0                // experiment: consumers free their groups without the dot products (WRONG results; the loader's own pace)
#endif
#ifndef RWKV_MG_NOSYNC
#define RWKV_MG_NOSYNC 0               // experiment: no chip-wide arrival polls (WRONG results; what the edges cost)
#endif
#ifndef RWKV_MG_NOEPI
#define RWKV_MG_NOEPI 0                // experiment: row epilogues skipped (WRONG results)
#endif
#ifndef RWKV_MG_NOPRE
#define RWKV_MG_NOPRE 0                // experiment: epilogue inputs not loaded (WRONG results; isolates their cost)
#endif
#ifndef RWKV_MG_THIN
#define RWKV_MG_THIN 0                 // older rule: pause only between poll success and gathered
#endif

enum { MG_FIRST = 0, MG_ATT = 1, MG_ATTOUT = 2, MG_FRK = 3, MG_FV = 4, MG_HEAD = 5 };

struct MegaSync {                      // device memory, zeroed at creation / after an error
    unsigned cnt[8][16];               // arrival counters, one 64-byte line per shard (workgroup b -> shard b & 7)
    unsigned fin[16];                  // arrivals at the token's last phase (returning atomic: the last one closes the token)
    unsigned epoch[16];                // tokens completed
    unsigned error[16];                // != 0: a wait gave up (results invalid)
};

struct MegaCtl {                       // LDS control block
    unsigned landed;                   // ring units whose DMA has completed            (loader -> consumers, monotonic)
    unsigned done;                     // consumer waves that finished a phase           (7 per phase, monotonic)
    unsigned staged;                   // prologue waves that staged their share         (4 per prologue)
    unsigned go;                       // prologues whose inputs are complete chip-wide  (wave 0 -> waves 1..3)
    unsigned meet;                     // prologue reduction rendezvous                  (4 per prologue)
    unsigned gathered;                 // prologue waves whose hand-off loads have landed (4 per prologue; the loader pauses in between)
    unsigned dead;                     // a wait gave up
    unsigned pad;
    unsigned freeq[MG_FQ];             // freeq[k % FQ] = k + 1: group k has been copied out of the ring
    unsigned gend[MG_FQ];              // loader's own: end unit of group k
    float bc[8];                       // scalars of the staged vectors: S_m at [m], amax_m at [4 + m]
    double lnst[2];                    // mean, rstd of the open LayerNorm site (att -> att_out, ffn_rk -> ffn_v)
    double wpd[MG_NC][8];              // per-wave partial tuples of a phase
    float wpf[MG_NC][4];
    unsigned wpi[MG_NC];
    unsigned pad2[9];
    unsigned long long ltl[16][2];     // timeline: the loader's stamps of phase q at [q % 16] (the loader makes no global stores: they would count in its vmcnt)
};
static_assert(sizeof(MegaCtl) % 16 == 0, "MegaCtl");

struct MegaOut {                       // cross-workgroup outputs of a phase, written out by the workgroup's last wave
    double x[MG_MAXROWS];              // att_out, ffn_v, first: new residual rows
    float B[3][MG_MAXROWS];            // the site they open
    float h[4 * MG_MAXROWS];           // ffn_rk: hidden units; att: ybuf at h[0..)
    float rg[MG_MAXROWS];              // receptance gate, ffn_rk -> ffn_v of the same workgroup (never leaves LDS)
};

template <int S> struct MegaLds {
    static constexpr int XQ = RED_BYTES;
    static constexpr int CTL = XQ + 4 * S * 3072;
    static constexpr int OUT = CTL + (int)sizeof(MegaCtl);
    static constexpr int RING = (OUT + (int)sizeof(MegaOut) + 1023) / 1024 * 1024;
    static constexpr int UNIT = S * 1024;
    static constexpr int NU = (160 * 1024 - RING) / UNIT;
    static constexpr int BYTES = RING + NU * UNIT;
    static constexpr int LAG = (63 - S + S - 1) / S;     // after vmcnt(63 - S) everything but the last LAG units has landed
};

struct MegaArgs {
    FirstArgs first;
    HeadArgs head;
    const AttArgs *att;                // device arrays, one entry per layer of this context
    const AttOutArgs *attout;
    const FfnRKArgs *frk;
    const FfnVArgs *fv;
    MegaSync *sync;
    unsigned *herr;                    // host-visible error word (mapped pinned memory)
    Ctl *ctl;
    unsigned long long *gen;
    unsigned gen_cap;
    int nl, has_head, with_argmax, D;
    unsigned long long *tl;            // optional: [grid][phases][MG_TL] wall-clock stamps
};

// The per-layer argument tables are written once at load time: read them through the constant address space, which makes the
// accesses scalar loads (a plain load of memory the kernel may also store to becomes a VECTOR load -- a vmcnt(0) drain in the
// loader, a cold vector miss at the head of every prologue)
template <class T> __device__ __forceinline__ const __attribute__((address_space(4))) T *as_const(const T *p)
{
    return (const __attribute__((address_space(4))) T *)(unsigned long long)p;
}
template <class T> __device__ __forceinline__ T load_const(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "argument blocks are made of 4- and 8-byte fields");
    constexpr int N = sizeof(T) / 4;
    const __attribute__((address_space(4))) unsigned *q = (const __attribute__((address_space(4))) unsigned *)(unsigned long long)p;
    unsigned tmp[N];
#pragma unroll
    for (int i = 0; i < N; i++) tmp[i] = q[i];
    T out;
    __builtin_memcpy(&out, tmp, sizeof(T));
    return out;
}
// ------------------------------------------------------------------------------------------ waits
__device__ __forceinline__ void mg_give_up(MegaCtl *mc, const MegaArgs &a)
{
    __hip_atomic_store(&mc->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&a.sync->error[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.herr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// LDS counter >= want (wrap-safe)
__device__ __forceinline__ void mg_wait_ge(const unsigned *p, unsigned want, MegaCtl *mc, const MegaArgs &a)
{
    for (int it = 0;; it++) {
        if ((int)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - want) >= 0) break;
        if ((it & 63) == 63 && (it >= MG_SPIN || __hip_atomic_load(&mc->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) { mg_give_up(mc, a); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void mg_wait_eq(const unsigned *p, unsigned want, MegaCtl *mc, const MegaArgs &a)
{
    for (int it = 0;; it++) {
        if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == want) break;
        if (it >= MG_SPIN || __hip_atomic_load(&mc->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { mg_give_up(mc, a); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// arrivals of shard s per phase in a grid of G workgroups
__device__ __forceinline__ unsigned mg_shard_n(unsigned G, unsigned s) { return (G + 7u - s) >> 3; }
// chip-wide: `nph` phases of this token are complete (one wave; lanes 0..7 watch one shard each, lane 8 the error word)
__device__ __forceinline__ void mg_poll(const MegaArgs &a, MegaCtl *mc, unsigned base_phases, unsigned nph, int lane)
{
    if (RWKV_MG_NOSYNC) return;
    const unsigned G = gridDim.x;
    const unsigned s = lane & 7;
    const unsigned want = (base_phases + nph) * mg_shard_n(G, s);
    const unsigned *p = lane == 8 ? &a.sync->error[0] : &a.sync->cnt[s][0];
    for (int it = 0;; it++) {
        unsigned v = 0u;
        if (lane < 9) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = lane == 8 ? v == 0u : (lane > 8 || (int)(v - want) >= 0);
        const bool err = lane == 8 && v != 0u;
        if (__builtin_amdgcn_ballot_w64(err) != 0ull) { __hip_atomic_store(&mc->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (it >= MG_SPIN) { mg_give_up(mc, a); break; }
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}
// 16-byte agent-scope (sc1) load: bypasses the L1, which another CU's stores never refresh
__device__ __forceinline__ u32x4 ld_sc16(const void *base, unsigned byte_off)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
}
template <class T> __device__ __forceinline__ T ld_sc(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void st_sc(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void mg_stamp(const MegaArgs &a, int nq, int q, int slot, int lane)
{
    if (a.tl) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = wall_clock64();
        if (lane == 0) a.tl[((size_t)blockIdx.x * nq + q) * MG_TL + slot] = t;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------ phase list
// q = 0: first; q = 1 + 4 li + (kind - 1): layer li; q = 1 + 4 nl: head
__device__ __forceinline__ int mg_nq(const MegaArgs &a) { return 1 + 4 * a.nl + (a.has_head ? 1 : 0); }
__device__ __forceinline__ void mg_decode(const MegaArgs &a, int q, int &kind, int &li)
{
    if (q == 0) { kind = MG_FIRST; li = 0; }
    else if (q <= 4 * a.nl) { kind = 1 + ((q - 1) & 3); li = (q - 1) >> 2; }
    else { kind = MG_HEAD; li = 0; }
}
// this workgroup's groups of a phase
__device__ __forceinline__ void mg_range(int kind, int D, int &g0, int &g1)
{
    const unsigned tot = kind == MG_ATTOUT ? (unsigned)(D + MG_AR - 1) / MG_AR : (kind == MG_HEAD ? (VOCAB + MG_HR - 1) / MG_HR : (unsigned)D);
    g0 = (int)((blockIdx.x * tot) / gridDim.x);        // < 2^32: grid <= 256, tot <= 12570
    g1 = (int)(((blockIdx.x + 1u) * tot) / gridDim.x);
}

// ------------------------------------------------------------------------------------------ loader
// One wave issues the chip's whole weight stream of this CU, so its instruction count IS the stream's ceiling (a wave issues one
// instruction per ~5 clocks; at 60 instructions per 4 KiB unit the stream stood at 13 KB/us per CU, under half of what the DMA
// path delivers).  Per unit: one s_waitcnt, one address add, one asm statement.  Per group: the bookkeeping.
template <int S> struct MegaLoader {
    static constexpr int NU = MegaLds<S>::NU;
    static constexpr int DEEP = (RWKV_MG_DEPTH < 63 ? RWKV_MG_DEPTH : 63) - S;                                   // vmcnt before a unit's issue
    static constexpr int THIN = RWKV_MG_THINQ > S ? (RWKV_MG_THINQ < 63 ? RWKV_MG_THINQ : 63) - S : 0;
    const MegaArgs &a;
    MegaCtl *mc;
    unsigned ring;           // LDS byte address of unit 0
    unsigned off[S];         // lane's byte offset in a row, per step (rows that are not whole KiB: the last pieces are clamped)
    bool whole;              // rows are whole KiB: a unit is one dma_unit
    unsigned issued = 0;     // units issued
    unsigned pub = 0;        // units published as landed
    unsigned k = 0, tail = 0;   // groups issued / groups known to be copied out
    unsigned tailu = 0;      // first unit still in use
    unsigned pos = 0;        // ring position of the next unit
    unsigned Pc = 0;         // phase the consumers were last seen in
    int lane;

    __device__ __forceinline__ MegaLoader(const MegaArgs &a_, MegaCtl *mc_, unsigned ring_, int lane_) : a(a_), mc(mc_), ring((unsigned)__builtin_amdgcn_readfirstlane((int)ring_)), lane(lane_)
    {
        const int chunks = a.D >> 4;
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
    // DMA instructions of this wave that have not completed: the wave's own vmcnt, read without waiting for it
    // (HW_REG_IB_STS: VM_CNT in bits 3:0, its two high bits in 23:22).  Loads complete in order and this wave issues no
    // other vector memory instruction, so everything but the last `vm` pieces is in LDS.
    __device__ __forceinline__ unsigned in_flight() const
    {
        const unsigned v = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 7);
        return (v & 0xfu) | ((v >> 18) & 0x30u);
    }
    __device__ __forceinline__ void poll_landed() { publish((issued * (unsigned)S - in_flight()) / (unsigned)S); }
    __device__ __forceinline__ void drain() { wait_vm<0>(); publish(issued); }
    __device__ __forceinline__ void lstamp(int q, int i)
    {
        if (a.tl) {
            const unsigned long long t = wall_clock64();
            if (lane == 0) mc->ltl[q & 15][i] = t;
        }
    }
    // Which groups have been copied out?  ONE LDS round trip answers for the next 32 groups (lane i looks at group tail + i);
    // the tail moves over every leading group that is free.  All loader state is wave-uniform and kept in SGPRs: every value
    // that comes back from LDS goes through readfirstlane / ballot / readlane before it touches that state.
    __device__ __forceinline__ void advance_tail()
    {
        const unsigned g = tail + (unsigned)(lane & 31);
        const unsigned f = __hip_atomic_load(&mc->freeq[g % MG_FQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned e = __hip_atomic_load(&mc->gend[g % MG_FQ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");      // compiler-level ordering is all LDS needs: a consumer's reads precede its flag in the LDS queue
        const bool ok = lane < 32 && (g - tail) < (k - tail) && f == g + 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        const unsigned n = (unsigned)__builtin_ctzll(~m);          // leading groups that are free (<= 32: bits 32.. are never set)
        if (n) {
            tailu = (unsigned)__builtin_amdgcn_readlane((int)e, (int)(n - 1u));
            tail += n;
        }
    }
    template <int R> __device__ __forceinline__ bool room() const { return !(issued + R - tailu > (unsigned)NU || k - tail >= (unsigned)MG_FQ); }
    __device__ __forceinline__ bool dead_or_late(int it)
    {
        if ((it & 63) == 63 && (it >= MG_SPIN || __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&mc->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))) { mg_give_up(mc, a); return true; }
        return false;
    }
    template <int R, bool THINM> __device__ __forceinline__ void issue(const uint8_t *src)
    {
        const size_t D = (size_t)a.D;
#pragma unroll
        for (int r = 0; r < R; r++) {
            wait_vm<THINM ? THIN : DEEP>();
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring + pos * (unsigned)(S * 1024)));
            if (whole) dma_unit<S>(src + r * D, dst);
            else {
#pragma unroll
                for (int s = 0; s < S; s++) dma_piece(src + r * D + (off[s] - off[0]), dst + s * 1024);
            }
            pos = pos + 1 == (unsigned)NU ? 0u : pos + 1;
        }
    }
    // R rows starting at `src` (this lane's first piece of row 0; rows D bytes apart) -> the next R units.  The loader never sits
    // in a long s_waitcnt: while it waits for room it keeps announcing what has landed -- the consumer that will free the
    // wanted units may be waiting for exactly that.
    template <int R> __device__ __forceinline__ void group(const uint8_t *src, unsigned q)
    {
        for (int it = 0; !room<R>(); it++) {
            advance_tail();
            if (room<R>()) break;
            poll_landed();
            if (dead_or_late(it)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        // where are the consumers?  staged = 4 P: they are in phase P (or in the prologue of P + 1); P only grows
        bool ahead = false;
        if (q > Pc) {
            for (int it = 0;; it++) {
                Pc = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&mc->staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) / MG_NP;
                ahead = q > Pc;
                if (!ahead || !RWKV_MG_PAUSE) break;
                const unsigned dn = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&mc->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const unsigned ga = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&mc->gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (!((int)(dn - (Pc + 1u) * MG_NC) >= 0 && (int)(ga - (Pc + 1u) * MG_NP) < 0)) break;     // not between "done" and "gathered"
                poll_landed();
                if (dead_or_late(it)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (lane == 0) mc->gend[k % MG_FQ] = issued + R;
        if (ahead) issue<R, true>(src); else issue<R, false>(src);
        issued += R;
        k++;
        poll_landed();
    }
    template <int R> __device__ __forceinline__ void phase(const uint8_t *w, int g0, int g1, int rows, int q)
    {
        const uint8_t *wl = w + off[0];
        lstamp(q, 0);
        for (int g = g0; g < g1; g++) {
            int row = g * R;
            if (row > rows - R) row = rows - R;     // the last group of a matrix may overlap the one before
            group<R>(wl + (size_t)row * a.D, (unsigned)q);
        }
        lstamp(q, 1);
    }
    __device__ __forceinline__ void run()
    {
        const int nq = mg_nq(a), D = a.D;
        for (int q = 1; q < nq; q++) {
            int kind, li, g0, g1;
            mg_decode(a, q, kind, li);
            mg_range(kind, D, g0, g1);
            switch (kind) {
            case MG_ATT: phase<3>(as_const(a.att + li)->w, g0, g1, 3 * D, q); break;
            case MG_ATTOUT: phase<MG_AR>(as_const(a.attout + li)->w, g0, g1, D, q); break;
            case MG_FRK: phase<5>(as_const(a.frk + li)->w, g0, g1, 5 * D, q); break;
            case MG_FV: phase<4>(as_const(a.fv + li)->w, g0, g1, 4 * D, q); break;
            default: phase<MG_HR>(a.head.w, g0, g1, (int)VOCAB, q); break;
            }
        }
        for (int it = 0; it < MG_SPIN && in_flight() != 0u; it++) { poll_landed(); __builtin_amdgcn_s_sleep(1); }
        drain();
    }
};

// ------------------------------------------------------------------------------------------ consumers
// position bookkeeping of the consumer waves: every wave walks the same phase list, so the unit / group numbers the loader
// will use are known without asking it
struct MegaPos { unsigned ubase = 0, kbase = 0; };

// the groups of one phase that belong to consumer wave cw: local group kl = cw, cw + 7, ...
template <int S, int R, int PAT, class Pre, class Epi>
__device__ __forceinline__ void mg_groups(const MegaArgs &a, int ng, const MegaPos &mp, const unsigned char *ring, MegaCtl *mc, const unsigned *xq,
                                          int lane, int cw, int chunks, int nq, int q, Pre pre, Epi epi)
{
    constexpr int NU = MegaLds<S>::NU;
    for (int kl = cw; kl < ng; kl += MG_NC) {
        const auto in = pre(kl);
        u32x4 w[R][S];
        const unsigned uend = mp.ubase + (unsigned)(kl + 1) * R;
        mg_wait_ge(&mc->landed, uend, mc, a);
        if (kl == 0) mg_stamp(a, nq, q, 5, lane);
        unsigned p0 = (uend - R) % (unsigned)NU;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(ring + (size_t)p0 * (S * 1024)) + lane;
#pragma unroll
            for (int s = 0; s < S; s++) w[r][s] = p[s * 64];
            p0 = p0 + 1 == (unsigned)NU ? 0u : p0 + 1;
        }
        const unsigned kg = mp.kbase + (unsigned)kl;
        if (lane == 0) __hip_atomic_store(&mc->freeq[kg % MG_FQ], kg + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned long long T[R];
        if (RWKV_MG_NODOT) {
#pragma unroll
            for (int r = 0; r < R; r++) T[r] = w[r][0][0];
        } else {
            group_dot<R, S, PAT, false>(w, xq, lane, T, nullptr, 0, chunks, false);
        }
        if (!RWKV_MG_NOEPI) epi(kl, T, in);
    }
}

// ---- prologues (waves 0..3 work, waves 4..6 wait for `staged`) ----
// np = prologues before this one (this is prologue np + 1); nph = phases of this token that must be complete
template <int NV, int S>
__device__ __forceinline__ void mg_site(const MegaArgs &a, const SiteStatic &st, const SiteDyn &dy, const double *x, double *red, unsigned *xq, MegaCtl *mc,
                                        unsigned base_phases, unsigned nph, unsigned np, int nq, int q, double (&Sm)[NV], float (&am)[NV])
{
    constexpr int NTP = MG_NP * 64, NQP = (S * 256 + NTP - 1) / NTP;
    const int D = a.D, nqd = D >> 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < MG_NP) {
        double tc[NV];
        float mcx[NV];
#pragma unroll
        for (int m = 0; m < NV; m++) { tc[m] = st.TC[m]; mcx[m] = st.maxC[m]; }
        f32x4 Cq[NQP][NV];
#pragma unroll
        for (int i = 0; i < NQP; i++) {
            const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
#pragma unroll
            for (int m = 0; m < NV; m++) Cq[i][m] = reinterpret_cast<const f32x4 *>(st.C + (size_t)m * D)[qc];
        }
        if (wave == 0) {
            mg_poll(a, mc, base_phases, nph, lane);
            if (lane == 0) __hip_atomic_store(&mc->go, np + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            mg_stamp(a, nq, q, 2, lane);
        } else {
            mg_wait_ge(&mc->go, np + 1u, mc, a);
        }
        asm volatile("" ::: "memory");
        // hand-off data: this thread's partial tuple, its quads of x and of B
        const int n_part = (int)gridDim.x;
        SiteTuple tup;
        {
            const unsigned i = (int)threadIdx.x < n_part ? threadIdx.x : 0u;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) tup.d[k2] = __builtin_bit_cast(f64x2, ld_sc16(dy.pd, i * 64u + k2 * 16u));
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
        // one reduction round over the tuples (site_reduce of kernels.hip.h, meeting on a monotonic LDS counter)
        float *redf = reinterpret_cast<float *>(red + NW * 8);
        const bool real = (int)threadIdx.x < n_part;
        double d[8];
        float f[4];
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) d[k2] = real ? tup.d[k2 >> 1][k2 & 1] : 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) f[k2] = real ? tup.f[k2] : 0.f;
        asm volatile("" : "+v"(d[0]));      // the tuple is here
        if (wave == 0) mg_stamp(a, nq, q, 3, lane);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++)
            if (k2 < 2 || (k2 - 2) % 3 < NV) d[k2] = wave_sum(d[k2]);
#pragma unroll
        for (int k2 = 0; k2 < 1 + NV; k2++) f[k2] = wave_max(f[k2]);
        if (lane == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) red[wave * 8 + k2] = d[k2];
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) redf[wave * 4 + k2] = f[k2];
            __hip_atomic_fetch_add(&mc->meet, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        mg_wait_ge(&mc->meet, (np + 1u) * MG_NP, mc, a);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            if (k2 < 2 || (k2 - 2) % 3 < NV) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < MG_NP; i++) s += red[i * 8 + k2];
                d[k2] = s;
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 1 + NV; k2++) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MG_NP; i++) s = fmaxf(s, redf[i * 4 + k2]);
            f[k2] = s;
        }
        SiteRed<NV> sr;
        sr.mean = d[0] * st.invD;
        const double var = (d[1] - d[0] * sr.mean) * st.invDm1;   // reference: (D-1), no epsilon (rwkv.cu:43-44,53)
        sr.rstd = rsqrt(var);
        const double mrs = sr.mean * sr.rstd;
#pragma unroll
        for (int m = 0; m < NV; m++) {
            sr.S[m] = sr.rstd * d[2 + m] - mrs * tc[m] + d[5 + m];
            sr.amax[m] = (mcx[m] * (float)(((double)f[0] + fabs(sr.mean)) * sr.rstd) + f[1 + m]) * 1.0001f;
        }
        // loads return in order: once the last one is here the hand-off has been gathered and the loader may go on
        asm volatile("" : "+v"(Bq[NQP - 1][NV - 1]));
        if (lane == 0) __hip_atomic_fetch_add(&mc->gathered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        site_stage<NV, NQP, S, NTP>(xl, Cq, Bq, sr, xq, nqd);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int m = 0; m < NV; m++) { mc->bc[m] = (float)sr.S[m]; mc->bc[4 + m] = sr.amax[m]; }
            mc->lnst[0] = sr.mean; mc->lnst[1] = sr.rstd;
        }
        if (lane == 0) __hip_atomic_fetch_add(&mc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    mg_wait_ge(&mc->staged, (np + 1u) * MG_NP, mc, a);
#pragma unroll
    for (int m = 0; m < NV; m++) { Sm[m] = (double)mc->bc[m]; am[m] = mc->bc[4 + m]; }
    if (wave == 0) mg_stamp(a, nq, q, 4, lane);
}

// plain-vector prologue (att_out: NVEC = 1, ybuf; ffn_v: NVEC = 4 quarter vectors of hbuf)
template <int NVEC, int S>
__device__ __forceinline__ void mg_vec(const MegaArgs &a, const float *vec, const double *partS, const float *partM, double *red, unsigned *xq, MegaCtl *mc,
                                       unsigned base_phases, unsigned nph, unsigned np, int nq, int q, float &Sf, float &amax)
{
    constexpr int XVD = xvd<S>();
    constexpr int NTP = MG_NP * 64, NQP = (S * 256 + NTP - 1) / NTP;
    const int D = a.D, nqd = D >> 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < MG_NP) {
        if (wave == 0) {
            mg_poll(a, mc, base_phases, nph, lane);
            if (lane == 0) __hip_atomic_store(&mc->go, np + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            mg_stamp(a, nq, q, 2, lane);
        } else {
            mg_wait_ge(&mc->go, np + 1u, mc, a);
        }
        asm volatile("" ::: "memory");
        const int n_part = (int)gridDim.x;
        const bool real = (int)threadIdx.x < n_part;
        double ps = ld_sc(partS + (real ? threadIdx.x : 0));
        float pm = ld_sc(partM + (real ? threadIdx.x : 0));
        float vl[NVEC][NQP][4];
#pragma unroll
        for (int v = 0; v < NVEC; v++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP, qc = qd < nqd ? qd : nqd - 1;
                const f32x4 t = __builtin_bit_cast(f32x4, ld_sc16(vec, ((unsigned)v * (unsigned)D + (unsigned)qc * 4u) * 4u));
                vl[v][i][0] = t[0]; vl[v][i][1] = t[1]; vl[v][i][2] = t[2]; vl[v][i][3] = t[3];
            }
        if (!real) { ps = 0.0; pm = 0.f; }
        asm volatile("" : "+v"(ps));
        if (wave == 0) mg_stamp(a, nq, q, 3, lane);
        float *redf = reinterpret_cast<float *>(red + RED_MAX);
        const double ws = wave_sum(ps);
        const float wm = wave_max(pm);
        if (lane == 0) {
            red[RED_OFFS + wave] = ws; redf[wave] = wm;
            __hip_atomic_fetch_add(&mc->meet, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        mg_wait_ge(&mc->meet, (np + 1u) * MG_NP, mc, a);
        double ts = 0.0; float tm = 0.f;
#pragma unroll
        for (int i = 0; i < MG_NP; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
        asm volatile("" : "+v"(vl[NVEC - 1][NQP - 1][3]));
        if (lane == 0) __hip_atomic_fetch_add(&mc->gathered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int v = 0; v < NVEC; v++)
#pragma unroll
            for (int i = 0; i < NQP; i++) {
                const int qd = threadIdx.x + i * NTP;
                if (qd < S * 256) stage_quad(xq + v * XVD, qd, vl[v][i], inv_scale(tm), qd < nqd);
            }
        if (threadIdx.x == 0) { mc->bc[0] = (float)ts; mc->bc[4] = tm; }
        if (lane == 0) __hip_atomic_fetch_add(&mc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    mg_wait_ge(&mc->staged, (np + 1u) * MG_NP, mc, a);
    Sf = mc->bc[0]; amax = mc->bc[4];
    if (wave == 0) mg_stamp(a, nq, q, 4, lane);
}

// row owner: emit B_m[local row] into LDS and accumulate the tuple share (site_emit of kernels.hip.h with an LDS destination);
// p = the channel's producer-side constants (SitePre<NV>::p), d / f = the tuple accumulators (SiteAcc layout)
template <int NV>
__device__ __forceinline__ void mg_emit(const f32x4 *p, MegaOut *mo, int li, double x, double prev, double (&d)[8], float (&f)[4])
{
    d[0] += x;
    d[1] += x * x;
    f[0] = fmaxf(f[0], (float)fabs(x) * 1.0000002f);
#pragma unroll
    for (int m = 0; m < NV; m++) {
        const float bl = p[(m * 5 + 0) >> 2][(m * 5 + 0) & 3], bp = p[(m * 5 + 1) >> 2][(m * 5 + 1) & 3];
        const float co = p[(m * 5 + 2) >> 2][(m * 5 + 2) & 3];
        const float bol = p[(m * 5 + 3) >> 2][(m * 5 + 3) & 3], bop = p[(m * 5 + 4) >> 2][(m * 5 + 4) & 3];
        const float b = (float)((double)bl + (double)bp * prev);
        mo->B[m][li] = b;
        d[2 + m] += (double)co * x;
        d[5 + m] += (double)bol + (double)bop * prev;
        f[1 + m] = fmaxf(f[1 + m], fabsf(b));
    }
}

// what a phase leaves behind, and the end of the phase.  Every consumer wave calls mg_finish once per phase with its partial
// (d[8], f[4]: a site tuple, or {sum, -; max} for the plain vectors, or the wave's best logit); the last wave to arrive
// writes the workgroup's tuple and its outputs and arrives at the chip-wide counter.
enum { OUT_NONE = 0, OUT_Y = 1, OUT_XB = 2, OUT_H = 3, OUT_BEST = 4 };
struct MegaOutDesc {
    int kind;                 // OUT_*
    int nv;                   // OUT_XB: vectors of the opened site
    int row0, nrows;          // first global row / count of the outputs
    double *x; float *B;      // OUT_XB
    float *vec;               // OUT_Y / OUT_H destination
    double *pd; float *pf;    // tuple destination (OUT_XB: [grid][8] / [grid][4]; OUT_Y / OUT_H: partS / partM [grid])
    float *blk_val; unsigned *blk_idx;
};
template <class F>
__device__ __forceinline__ void mg_finish(const MegaArgs &a, MegaCtl *mc, MegaOut *mo, const MegaOutDesc &od, double (&d)[8], float (&f)[4], unsigned bi,
                                          unsigned nph_done /* phases complete once this one is */, bool last_phase, unsigned epoch, int cw, int lane, int nq, int q, F closing)
{
    const int D = a.D;
    // the wave's partial (OUT_BEST: f[0] = best logit, bi its index -- every lane holds the same pair, the sums are wave-uniform)
    if (od.kind == OUT_XB) {
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) d[k2] = wave_sum(d[k2]);
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) f[k2] = wave_max(f[k2]);
    } else if (od.kind != OUT_BEST) {
        d[0] = wave_sum(d[0]);
        f[0] = wave_max(f[0]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) mc->wpd[cw][k2] = d[k2];
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) mc->wpf[cw][k2] = f[k2];
        mc->wpi[cw] = bi;
    }
    // this wave's own global stores (recurrent state: read by this workgroup's successor in the NEXT token) need no drain
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&mc->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    if (old + 1u != nph_done * MG_NC) return;
    // ---- last wave of the workgroup ----
    const unsigned b = blockIdx.x;
    if (od.kind == OUT_XB) {
        if (lane < 8) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < MG_NC; i++) t += mc->wpd[i][lane];
            st_sc(od.pd + (size_t)b * 8 + lane, t);
        } else if (lane < 12) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < MG_NC; i++) t = fmaxf(t, mc->wpf[i][lane - 8]);
            st_sc(od.pf + (size_t)b * 4 + (lane - 8), t);
        }
        for (int i = lane; i < od.nrows; i += 64) {
            st_sc(od.x + od.row0 + i, mo->x[i]);
            for (int m = 0; m < od.nv; m++) st_sc(od.B + (size_t)m * D + od.row0 + i, mo->B[m][i]);
        }
    } else if (od.kind == OUT_Y || od.kind == OUT_H) {
        if (lane == 0) {
            double t = 0.0; float tm = 0.f;
#pragma unroll
            for (int i = 0; i < MG_NC; i++) { t += mc->wpd[i][0]; tm = fmaxf(tm, mc->wpf[i][0]); }
            st_sc(od.pd + b, t);
            st_sc(od.pf + b, tm);
        }
        for (int i = lane; i < od.nrows; i += 64) st_sc(od.vec + od.row0 + i, mo->h[i]);
    } else if (od.kind == OUT_BEST) {
        if (lane == 0) {
            float best = mc->wpf[0][0]; unsigned besti = mc->wpi[0];
            for (int i = 1; i < MG_NC; i++) {
                const float v = mc->wpf[i][0]; const unsigned id = mc->wpi[i];
                if (v > best || (v == best && id < besti)) { best = v; besti = id; }
            }
            st_sc(od.blk_val + b, best);
            st_sc(od.blk_idx + b, besti);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores are acknowledged before the arrival is visible
    mg_stamp(a, nq, q, 1, lane);
    if (a.tl && lane < 2) a.tl[((size_t)blockIdx.x * nq + q) * MG_TL + 6 + lane] = mc->ltl[q & 15][lane];
    if (!last_phase) {
        if (lane == 0) __hip_atomic_fetch_add(&a.sync->cnt[b & 7u][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        unsigned o = 0;
        if (lane == 0) o = __hip_atomic_fetch_add(&a.sync->fin[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o = (unsigned)__builtin_amdgcn_readfirstlane((int)o);
        if (o + 1u == (epoch + 1u) * gridDim.x) {
            closing();
            if (lane == 0) a.sync->epoch[0] = epoch + 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------ the token
template <int S>
__global__ __launch_bounds__(NT) void k_token(MegaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using LY = MegaLds<S>;
    double *red = reinterpret_cast<double *>(smem);
    unsigned *xq = reinterpret_cast<unsigned *>(smem + LY::XQ);
    MegaCtl *mc = reinterpret_cast<MegaCtl *>(smem + LY::CTL);
    MegaOut *mo = reinterpret_cast<MegaOut *>(smem + LY::OUT);
    unsigned char *ring = smem + LY::RING;
    const int D = a.D, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunks = D >> 4;
    const int nq = mg_nq(a);
    const unsigned G = gridDim.x, b = blockIdx.x;

    if (threadIdx.x < sizeof(MegaCtl) / 4) reinterpret_cast<unsigned *>(mc)[threadIdx.x] = 0u;
    const unsigned epoch = a.sync->epoch[0];
    const unsigned base_phases = epoch * (unsigned)(nq - 1);      // arrivals per workgroup before this token
    const size_t so = (size_t)a.ctl->slot * a.first.slot_stride;
    const unsigned long long token = a.ctl->token;
    const unsigned out_row = a.ctl->out_row;
    __syncthreads();       // the only workgroup barrier: the control block is zero

    if (wave == MG_NC) {
        MegaLoader<S> ld(a, mc, lds_addr(ring), lane);
        ld.run();
        return;
    }
    const int cw = wave;
    MegaPos mp;
    unsigned np = 0;           // prologues done
    auto no_closing = [] {};

    // ---------------- phase 0: residual vector of the token + the first layer's ln1 site (k_first) ----------------
    {
        int g0, g1;
        mg_range(MG_FIRST, D, g0, g1);
        SiteAcc<3> acc;
        acc.clear();
        if (cw == 0) {
            const FirstArgs &fa = a.first;
            double mean = 0.0, rstd = 1.0;
            const float *row = fa.from_token ? fa.embed + (size_t)token * D : nullptr;
            if (fa.from_token) {
                double s0 = 0.0, s1 = 0.0;
                for (int c = lane; c < (D >> 2); c += 64) {
                    const f32x4 v = reinterpret_cast<const f32x4 *>(row)[c];
#pragma unroll
                    for (int e = 0; e < 4; e++) { const double t = (double)v[e]; s0 += t; s1 += t * t; }
                }
                s0 = wave_sum(s0); s1 = wave_sum(s1);
                mean = s0 / (double)D;
                rstd = 1.0 / sqrt((s1 - s0 * mean) / (double)(D - 1));
            }
            for (int j = g0 + lane; j < g1; j += 64) {
                double x;
                if (fa.from_token) x = fa.ln[j] * (((double)row[j] - mean) * rstd) + fa.ln[D + j];
                else x = fa.x_in[j];
                SitePre<3> pre;
                site_prefetch<3>(fa.st, j, pre);
                mo->x[j - g0] = x;
                mg_emit<3>(pre.p, mo, j - g0, x, fa.sxy[so + j], acc.d, acc.f);
            }
        }
        MegaOutDesc od{};
        od.kind = OUT_XB; od.nv = 3; od.row0 = g0; od.nrows = g1 - g0; od.x = a.first.x; od.B = a.first.dy.B; od.pd = a.first.dy.pd; od.pf = a.first.dy.pf;
        mg_finish(a, mc, mo, od, acc.d, acc.f, 0u, 1u, nq == 1, epoch, cw, lane, nq, 0, no_closing);
    }

    // ---------------- layers ----------------
    for (int li = 0; li < a.nl; li++) {
        const int qb = 1 + 4 * li;
        // ---- att: ln1 site -> K, V, R rows -> WKV (k_att) ----
        {
            const AttArgs aa = load_const(a.att + li);
            const int q = qb;
            int g0, g1;
            mg_range(MG_ATT, D, g0, g1);
            double Sm[3]; float am[3];
            mg_site<3, S>(a, aa.st, aa.dy, aa.x, red, xq, mc, base_phases, (unsigned)q, np, nq, q, Sm, am);
            np++;
            const double sc0 = scale_of(am[0]), sc1 = scale_of(am[1]), sc2 = scale_of(am[2]);
            const float S0 = (float)Sm[0], S1 = (float)Sm[1], S2 = (float)Sm[2];
            double part = 0.0;
            float pmax = 0.f;
            auto pre = [&](int kl) {
                const int g = g0 + kl;
                AttIn in;
                if (RWKV_MG_NOPRE) { in = AttIn{}; in.bb = 1.0; return in; }
#pragma unroll
                for (int m = 0; m < 3; m++) in.rs[m] = aa.rs[g * 3 + m];
                in.aa = aa.saa[so + g]; in.bb = aa.sbb[so + g]; in.uw = aa.uw[g]; in.ew = aa.ew[g];
                in.ra = aa.r_att[g]; in.oa = aa.o_att[g];
                return in;
            };
            auto epi = [&](int kl, const unsigned long long (&T)[3], const AttIn &in) {
                if (lane == 0) {
                    const int g = g0 + kl;
                    const float k = row_value(T[0], in.rs[0], sc0) + S0, v = row_value(T[1], in.rs[1], sc1) + S1;
                    const float r = row_value(T[2], in.rs[2], sc2) + S2;
                    const double vv = (double)v;
                    const double e1 = exp(in.uw + (double)k);
                    double y = (in.aa + e1 * vv) / (in.bb + e1);
                    y = (1.0 / (1.0 + (double)expf(-r))) * y;       // rwkv.cu:250
                    const double ek = exp((double)k);
                    aa.saa[so + g] = (in.aa + ek * vv) * in.ew;
                    aa.sbb[so + g] = (in.bb + ek) * in.ew;
                    const float yf = (float)y;
                    const float ys = yf * in.ra;
                    mo->h[kl] = ys;
                    part += (double)(yf * in.oa);
                    pmax = fmaxf(pmax, fabsf(ys));
                }
            };
            if (cw == 0) mg_stamp(a, nq, q, 0, lane);
            mg_groups<S, 3, PAT_PER_ROW>(a, g1 - g0, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 3u * (unsigned)(g1 - g0); mp.kbase += (unsigned)(g1 - g0);
            double d[8] = {part, 0, 0, 0, 0, 0, 0, 0};
            float f[4] = {pmax, 0.f, 0.f, 0.f};
            MegaOutDesc od{};
            od.kind = OUT_Y; od.row0 = g0; od.nrows = g1 - g0; od.vec = aa.ybuf; od.pd = aa.partS; od.pf = aa.partM;
            mg_finish(a, mc, mo, od, d, f, 0u, (unsigned)q + 1u, false, epoch, cw, lane, nq, q, no_closing);
        }
        // ---- att_out: GEMV + residual; commits state xy; opens the ln2 site (k_attout) ----
        {
            constexpr int R = MG_AR;
            const AttOutArgs ao = load_const(a.attout + li);
            const int q = qb + 1;
            int g0, g1;
            mg_range(MG_ATTOUT, D, g0, g1);
            float Sf, amax;
            mg_vec<1, S>(a, ao.ybuf, ao.partS, ao.partM, red, xq, mc, base_phases, (unsigned)q, np, nq, q, Sf, amax);
            np++;
            const double sc = scale_of(amax);
            const double mean1 = mc->lnst[0], rstd1 = mc->lnst[1];
            const int rowlo = g0 * R;
            SiteAcc<2> acc;
            acc.clear();
            auto pre = [&](int kl) {
                AttOutIn<R> in;
                int row0 = (g0 + kl) * R;
                in.shift = (row0 > D - R) ? row0 - (D - R) : 0;
                row0 -= in.shift;
                in.mi = row0 + (lane < R ? lane : 0);
                if (RWKV_MG_NOPRE) { in.rsum = 0; in.xold = in.lw = in.lb = in.prev2 = 0.0; in.pre = SitePre<2>{}; return in; }
                in.rsum = ao.rs[in.mi];
                in.xold = ld_sc(ao.x + in.mi); in.lw = ao.lnw[in.mi]; in.lb = ao.lnb[in.mi]; in.prev2 = ao.sdd[so + in.mi];
                site_prefetch<2>(ao.st, in.mi, in.pre);
                return in;
            };
            auto epi = [&](int, const unsigned long long (&T)[R], const AttOutIn<R> &in) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (lane == r && r >= in.shift) {
                        const float accf = (float)in.xold + (row_value(T[r], in.rsum, sc) + Sf);   // rwkv.cu:548
                        const double xnew = (double)accf;                                           // :553
                        mo->x[in.mi - rowlo] = xnew;
                        ao.sxy[so + in.mi] = in.lw * ((in.xold - mean1) * rstd1) + in.lb;           // :385
                        mg_emit<2>(in.pre.p, mo, in.mi - rowlo, xnew, in.prev2, acc.d, acc.f);
                    }
                }
            };
            if (cw == 0) mg_stamp(a, nq, q, 0, lane);
            mg_groups<S, R, PAT_SHARED>(a, g1 - g0, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += (unsigned)R * (unsigned)(g1 - g0); mp.kbase += (unsigned)(g1 - g0);
            MegaOutDesc od{};
            const int rowhi = g1 * R < D ? g1 * R : D;
            od.kind = OUT_XB; od.nv = 2; od.row0 = rowlo; od.nrows = rowhi - rowlo; od.x = ao.x; od.B = ao.dy.B; od.pd = ao.dy.pd; od.pf = ao.dy.pf;
            mg_finish(a, mc, mo, od, acc.d, acc.f, 0u, (unsigned)q + 1u, false, epoch, cw, lane, nq, q, no_closing);
        }
        // ---- ffn_rk: ln2 site -> ffn_r + sigmoid, ffn_k + relu^2 (k_ffn_rk) ----
        {
            const FfnRKArgs fa = load_const(a.frk + li);
            const int q = qb + 2;
            int g0, g1;
            mg_range(MG_FRK, D, g0, g1);
            double Sm[2]; float am[2];
            mg_site<2, S>(a, fa.st, fa.dy, fa.x, red, xq, mc, base_phases, (unsigned)q, np, nq, q, Sm, am);
            np++;
            const double sck = scale_of(am[0]), scr = scale_of(am[1]);
            const float Sk = (float)Sm[0], Sr = (float)Sm[1];
            double part = 0.0;
            float pmax = 0.f;
            auto pre = [&](int kl) {
                const int g = g0 + kl;
                FfnRKIn in;
                if (RWKV_MG_NOPRE) { in.rsum = 0; in.rq = in.oq = 0.f; return in; }
                in.rsum = fa.rs[g * 5 + (lane < 5 ? lane : 0)];
                const int kk = 4 * g + (lane < 4 ? lane : 0);
                in.rq = fa.r_fv[kk]; in.oq = fa.o_fv[kk];
                return in;
            };
            auto epi = [&](int kl, const unsigned long long (&T)[5], const FfnRKIn &in) {
                float val = 0.f;
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const float vr = row_value(T[r], in.rsum, r < 4 ? sck : scr) + (r < 4 ? Sk : Sr);
                    val = lane == r ? vr : val;
                }
                if (lane < 4) {
                    float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
                    h = h * h;
                    const float hs = h * in.rq;
                    mo->h[4 * kl + lane] = hs;
                    part += (double)(h * in.oq);
                    pmax = fmaxf(pmax, fabsf(hs));
                } else if (lane == 4) {
                    mo->rg[kl] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
                }
            };
            if (cw == 0) mg_stamp(a, nq, q, 0, lane);
            mg_groups<S, 5, PAT_FFN_RK>(a, g1 - g0, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 5u * (unsigned)(g1 - g0); mp.kbase += (unsigned)(g1 - g0);
            double d[8] = {part, 0, 0, 0, 0, 0, 0, 0};
            float f[4] = {pmax, 0.f, 0.f, 0.f};
            MegaOutDesc od{};
            od.kind = OUT_H; od.row0 = 4 * g0; od.nrows = 4 * (g1 - g0); od.vec = fa.hbuf; od.pd = fa.partS; od.pf = fa.partM;
            mg_finish(a, mc, mo, od, d, f, 0u, (unsigned)q + 1u, false, epoch, cw, lane, nq, q, no_closing);
        }
        // ---- ffn_v: GEMV, x += v * sigmoid(r); commits state dd; opens the next site (k_ffnv) ----
        {
            const FfnVArgs fv = load_const(a.fv + li);
            const int q = qb + 3;
            const bool next_att = li + 1 < a.nl;      // else: ln_out -> head (1 vector; nobody reads it on a non-final stage)
            int g0, g1;
            mg_range(MG_FV, D, g0, g1);
            float Sf, amax;
            mg_vec<4, S>(a, fv.hbuf, fv.partS, fv.partM, red, xq, mc, base_phases, (unsigned)q, np, nq, q, Sf, amax);
            np++;
            const double sc = scale_of(amax);
            const double mean2 = mc->lnst[0], rstd2 = mc->lnst[1];
            SiteAcc<3> acc;
            acc.clear();
            auto pre = [&](int kl) {
                const int g = g0 + kl;
                FfnVIn<3> in;
                if (RWKV_MG_NOPRE) { in.rsum = 0; in.xold = in.lw = in.lb = in.prevn = 0.0; in.rg = 0.f; in.pre = SitePre<3>{}; return in; }
                in.rsum = fv.rs[g];
                in.xold = ld_sc(fv.x + g); in.lw = fv.lnw[g]; in.lb = fv.lnb[g];
                in.prevn = next_att ? fv.sprev[so + g] : 0.0;
                in.rg = mo->rg[kl];
                // the opened site has 3 vectors (16 floats per channel) or 1 (8 floats): both tables are read as the wider
                // struct's prefix, the 1-vector case uses p[0], p[1]
                if (next_att) site_prefetch<3>(fv.st, g, in.pre);
                else {
                    SitePre<1> p1;
                    site_prefetch<1>(fv.st, g, p1);
                    in.pre.p[0] = p1.p[0]; in.pre.p[1] = p1.p[1];
                }
                return in;
            };
            auto epi = [&](int kl, const unsigned long long (&T)[4], const FfnVIn<3> &in) {
                if (lane == 0) {
                    const int g = g0 + kl;
                    const float v = row_value((T[0] + T[1]) + (T[2] + T[3]), in.rsum, sc) + Sf;
                    const double xnew = in.xold + (double)(v * in.rg);               // rwkv.cu:407
                    mo->x[kl] = xnew;
                    fv.sdd[so + g] = in.lw * ((in.xold - mean2) * rstd2) + in.lb;    // :344
                    if (next_att) mg_emit<3>(in.pre.p, mo, kl, xnew, in.prevn, acc.d, acc.f);
                    else mg_emit<1>(in.pre.p, mo, kl, xnew, 0.0, acc.d, acc.f);
                }
            };
            if (cw == 0) mg_stamp(a, nq, q, 0, lane);
            mg_groups<S, 4, PAT_PER_ROW>(a, g1 - g0, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 4u * (unsigned)(g1 - g0); mp.kbase += (unsigned)(g1 - g0);
            MegaOutDesc od{};
            od.kind = OUT_XB; od.nv = next_att ? 3 : 1; od.row0 = g0; od.nrows = g1 - g0; od.x = fv.x; od.B = fv.dy.B; od.pd = fv.dy.pd; od.pf = fv.dy.pf;
            mg_finish(a, mc, mo, od, acc.d, acc.f, 0u, (unsigned)q + 1u, q == nq - 1, epoch, cw, lane, nq, q, no_closing);
        }
    }

    // ---------------- head: ln_out site -> logits (+ greedy pick) (k_head, k_argmax_finish) ----------------
    if (a.has_head) {
        constexpr int R = MG_HR;
        const HeadArgs &ha = a.head;
        const int q = nq - 1;
        const int V = (int)VOCAB;
        int g0, g1;
        mg_range(MG_HEAD, D, g0, g1);
        float *lg = ha.logits + (size_t)out_row * V;
        double Sm[1]; float am[1];
        mg_site<1, S>(a, ha.st, ha.dy, ha.x, red, xq, mc, base_phases, (unsigned)q, np, nq, q, Sm, am);
        np++;
        const float Sf = (float)Sm[0];
        const double sc = scale_of(am[0]);
        float best = -INFINITY;
        unsigned besti = 0xffffffffu;
        auto pre = [&](int kl) {
            HeadIn<R> in;
            in.row0 = (g0 + kl) * R;
            in.shift = (in.row0 > V - R) ? in.row0 - (V - R) : 0;
            in.row0 -= in.shift;
#pragma unroll
            for (int r = 0; r < R; r++) in.rsr[r] = RWKV_MG_NOPRE ? 0u : ha.rs[in.row0 + r];
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
        if (cw == 0) mg_stamp(a, nq, q, 0, lane);
        mg_groups<S, R, PAT_SHARED>(a, g1 - g0, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
        double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float f[4] = {best, 0.f, 0.f, 0.f};
        MegaOutDesc od{};
        od.kind = OUT_BEST; od.blk_val = ha.blk_val; od.blk_idx = ha.blk_idx;
        // the workgroup that arrives last finishes the greedy pick (k_argmax_finish): argmax over the workgroups' partials
        // (ties -> lowest id), fed back as the next token
        auto closing = [&] {
            if (!a.with_argmax) return;
            float bv = -INFINITY;
            unsigned bx = 0xffffffffu;
            for (unsigned i = lane; i < G; i += 64) {
                const float v = ld_sc(ha.blk_val + i);
                const unsigned id = ld_sc(ha.blk_idx + i);
                if (v > bv || (v == bv && id < bx)) { bv = v; bx = id; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ov = __shfl_xor(bv, m, 64);
                const unsigned oi = __shfl_xor(bx, m, 64);
                if (ov > bv || (ov == bv && oi < bx)) { bv = ov; bx = oi; }
            }
            if (lane == 0) {
                if (bx >= VOCAB) bx = 0u;
                const unsigned st = a.ctl->step;
                if (st < a.gen_cap) a.gen[st] = bx;
                a.ctl->token = bx;
                a.ctl->step = st + 1;
            }
        };
        mg_finish(a, mc, mo, od, d, f, besti, (unsigned)q + 1u, true, epoch, cw, lane, nq, q, closing);
    }
}

} // namespace rwkvk
