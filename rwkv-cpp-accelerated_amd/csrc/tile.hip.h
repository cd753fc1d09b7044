// tile.hip.h -- TILE-FORM decode kernels (round 5): the four per-layer kernels of a token on a TILE image of the matrices instead of the
// row-per-output re-tiling of kernels.hip.h.  Built for k_ffn_rk first -- the gate the round-4 review set ("decode on the MFMA B-operand
// image" against the row-form ring kernel on one box) --, then for all four classes; DESIGN.md 4.7 has the measurements, including those
// that decide which class runs in which form at which width (engine.hip tile_cfg_for / rwkv_ctx::tile).
//
// The row-form decode kernels stream a matrix re-tiled to row-per-output order: a wave owns whole rows, a lane 16 bytes of a row per
// step, every row ends in three 6-step DPP reductions and an epilogue on one lane.  These kernels stream the image k_bimage (seq.hip.h)
// makes: [TH-row tile][k-fragment of 1 KiB][lane][16 B], lane = TH * (piece along k) + row, bytes SIGNED (u - 128).  TH = 16 is the
// chunk path's MFMA B-operand image itself (4096 channels: one tile per class and workgroup, 64 inputs per fragment); TH = 4 is the
// same layout for widths where 16 channels per workgroup do not divide (5120: five tiles per class, 256 inputs per fragment; 2048: two).
//   * a wave instruction's 1 KiB is 1024 / TH inputs of TH rows: lane l accumulates row l % TH over piece l / TH of every fragment it
//     meets -- TH = 16: no cross-lane reduction per row at all, TH = 4: two DPP steps --; the pieces of a row and the k-ranges of
//     different waves meet as exact integer adds in LDS (ds_add_u32: the sums are integers, so the order they arrive in changes nothing);
//   * the ring unit is S KiB = S fragments of ONE tile (contiguous in the image: one dma_unit), units go to the 7 consumer waves round
//     robin (or in runs, tile_run), every unit is freed as soon as its S ds_read_b128 are in the LDS queue: fine-grained turnover, all
//     waves busy from the first unit on (row form: 5 groups of 20 KiB in the ring for 7 waves);
//   * the wave whose add completes a tile runs that tile's TH epilogues side by side on TH lanes;
//   * activations: the same 23-bit fixed point, limbs stored SIGNED (limb - 128) so that v_dot4_i32_i8 multiplies the image's bytes as
//     they are: sum u l = dot + 128 rowsum(u) + 128 sum(l) - 16384 N, folded into one constant per vector (cA) and the row-sum
//     coefficient 4227200 the chunk path uses (seq.hip.h SEQ_CU).  Every row value is the SAME exact integer as in row form: logits and
//     state are bit-identical to the row-form kernels' for every mix of forms (tests/test_engine_gpu.py);
//   * staged vector layout [16-byte piece along k][limb][16 B]: a lane's operand is one ds_read_b128, TH lanes share an address (broadcast).
// Integer contraction, since round 6 on the matrix cores: a 16-row fragment IS v_mfma_i32_16x16x64_i8's B operand as it lies in the ring, with the staged
// vector as the A operand (its three limbs = rows 0..2 of 16); a 4-row fragment is the B operand of 16 virtual rows (row, piece % 4) against an A whose
// twelve rows are (limb, piece % 4), the diagonal blocks of the product being what belongs together (tile_consume).  Per fragment one 16-byte LDS read and
// one instruction where the VALU form (v_dot4_i32_i8, kept behind RWKV_TILE_MFMA / RWKV_TILE_MFMA4 = 0) needs three reads and twelve dot instructions.
// Not a reshaping of the GEMV into a GEMM: a launch streams the same bytes once and stays HBM-bound; the consumers leave the LDS pipe to the DMA and
// follow the stream more closely (+1.2 % on the 7B token, +2.8 % at 14B, +1.7 % at 1B5).
#pragma once
#include "kernels.hip.h"

namespace rwkvk {

constexpr double TILE_CU = 4227200.0;          // 128 * (1 + 256 + 65536) - 2^22: coefficient of the unsigned row sum (= seq.hip.h SEQ_CU)
constexpr double TILE_CN = -1077952512.0;      // -16384 * 65793: coefficient of N (elements of the vector) beside 128 * sum_j (q_j + 2^22)

// TUNING builds (tools/stream_profile.py; never the shipped library): -DRWKV_TL_STREAM=1 the loader stamps the stream's progress (slot 1 behind the
// order barrier, 3 / 4 / 5 a quarter / half / three quarters of its units requested) and every consumer wave records the time it waited for
// units (slot 2); =2 every wave stamps its arrival at and departure from the order barrier (slots 1 and 2; the loader 3 and 1)
#if RWKV_TL_STREAM == 2
#define TL_BAR(tl, ph) tl_stamp(tl, ph)
#else
#define TL_BAR(tl, ph)
#endif
// Tiles are multiplied on the matrix cores (1; round 6, profiles/r06/mfma_consumer_ab.txt: 16-row tiles 614.4 -> 621.8 tokens/s at 7B; 4-row tiles
// 355.4 -> 365.4 at 14B, 1351 -> 1374 at 1B5) or with v_dot4 on the VALU (0: round 5's form).  Same exact integers either way; tile form == row form
// bit for bit is a GPU test (tests/test_engine_gpu.py).
#ifndef RWKV_TILE_MFMA           // 16-row tiles
#define RWKV_TILE_MFMA 1
#endif
#ifndef RWKV_TILE_MFMA4          // 4-row tiles
#define RWKV_TILE_MFMA4 1
#endif
constexpr bool TILE_MFMA = RWKV_TILE_MFMA != 0, TILE_MFMA4 = RWKV_TILE_MFMA4 != 0;
typedef int tile_i32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE_NWP = NT / 2 / 64;          // prologue waves (0 .. 3)
constexpr int TILE_NSTASH = NC - TILE_NWP;     // consumer waves that fetch the epilogues' inputs meanwhile (4 .. 6)
struct TileCtl {                 // LDS control block of a tile-form ring
    unsigned staged;             // waves that have done their part of what every consumer needs before its first unit: the TILE_NWP prologue waves (their
                                 // share of the vectors) + the TILE_NSTASH waves that fetch the epilogues' inputs into LDS
    unsigned landed;             // ring units whose DMA has completed (loader -> consumers, monotonic)
    unsigned freed[8];           // freed[w]: units consumer wave w has copied out of the ring (in ITS order: runs of RUN units, run r belongs to wave r % 7)
    unsigned tcnt[32];           // per tile of the workgroup: units already added into tsum
    unsigned done[8];            // k_att_t: per channel block of the workgroup, tiles (K, V, R) whose values are in
    unsigned pad[2];
    unsigned long long sq[4];    // per staged vector: sum of (q_j + 2^22) over its elements (one ds_add_u64 per staging wave)
    unsigned dump[64];           // where lanes 1..63 of the loader put their copy of `landed` (a store by ALL lanes needs no exec juggling: two instructions)
};
static_assert(sizeof(TileCtl) % 16 == 0, "ring behind the control block stays 16-byte aligned");

// stage 4 consecutive elements (quad qd) of a vector in TILE layout: dword dw of (k-block kb, quarter qq, limb b) at
// xq[kb * 48 + qq * 12 + b * 4 + dw], bytes = limb - 128.  Returns the sum of the four 23-bit values (q + 2^22) for the cA constant.
__device__ __forceinline__ unsigned stage_quad_t(unsigned *xq, int qd, const float (&xr)[4], float inv_s)
{
    unsigned t[4];
#pragma unroll
    for (int e = 0; e < 4; e++) t[e] = __float_as_uint(fmaf(xr[e], inv_s, QMAGIC));
    const unsigned p01 = __builtin_amdgcn_perm(t[1], t[0], 0x05010400u);
    const unsigned p23 = __builtin_amdgcn_perm(t[3], t[2], 0x05010400u);
    const unsigned h01 = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u);
    const unsigned h23 = __builtin_amdgcn_perm(t[3], t[2], 0x0c0c0602u);
    const unsigned d0 = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    const unsigned d1 = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
    const unsigned d2 = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
    unsigned *p = xq + (qd >> 4) * 48 + ((qd >> 2) & 3) * 12 + (qd & 3);
    p[0] = d0 ^ 0x80808080u;
    p[4] = d1 ^ 0x80808080u;
    p[8] = d2 ^ 0x80808080u;
    return (t[0] & 0x7fffffu) + (t[1] & 0x7fffffu) + (t[2] & 0x7fffffu) + (t[3] & 0x7fffffu);
}

// LayerNorm-site prologue of a tile-form kernel, run by waves 0..3 (the other consumer waves and the loader meet them at the order
// barrier inside); same ownership split as ring_site (kernels.hip.h).  SD = ceil(D / 1024).
template <int NV, int SD>
__device__ __forceinline__ void tile_site(const SiteStatic &st, const SiteDyn &dy, const double *x, int D, double *red, unsigned *xq, int xvd_t,
                                          bool publish_stats, TileCtl *tc, unsigned long long *tl)
{
    constexpr int NTP = NT / 2, NWP = NTP / 64, NQP = (SD * 256 + NTP - 1) / NTP;
    const int nqd = D >> 2;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    double tcs[NV];
    float mc[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) { tcs[m] = st.TC[m]; mc[m] = st.maxC[m]; }
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
    TL_BAR(tl, 2);
    SiteRed<NV> r;
    site_reduce<NV, NWP>(st, dy, tup, D, red, r, tcs, mc, tl, spin);
    if (publish_stats && blockIdx.x == 0 && threadIdx.x == 0) { dy.lnstat[0] = r.mean; dy.lnstat[1] = r.rstd; }
    tl_stamp(tl, 4);
    const float rstdf = (float)r.rstd;
    unsigned long long sq[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) sq[m] = 0ull;
#pragma unroll
    for (int i = 0; i < NQP; i++) {
        const int qd = threadIdx.x + i * NTP;
        if (qd < nqd) {
            float xh[4];
#pragma unroll
            for (int e = 0; e < 4; e++) xh[e] = (float)(xl[i][e] - r.mean) * rstdf;
#pragma unroll
            for (int m = 0; m < NV; m++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaf(Cq[i][m][e], xh[e], Bq[i][m][e]);
                sq[m] += stage_quad_t(xq + m * xvd_t, qd, v, inv_scale(r.amax[m]));
            }
        }
    }
    // one LDS atomic per wave and vector (64 lanes on one address would serialise in the LDS pipe under the DMA stream): the thread sums
    // (< 2^27 each) are folded on the DPP network in two 32-bit halves that cannot overflow
#pragma unroll
    for (int m = 0; m < NV; m++) {
        const unsigned lo = wave_sum_dpp((unsigned)sq[m] & 0xffffu), hi = wave_sum_dpp((unsigned)(sq[m] >> 16));
        if ((threadIdx.x & 63) == 0)
            __hip_atomic_fetch_add(&tc->sq[m], (unsigned long long)lo + ((unsigned long long)hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int m = 0; m < NV; m++) { bc[m] = (float)r.S[m]; bc[4 + m] = r.amax[m]; }
    }
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&tc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// S KiB from ONE wave-uniform address (SGPR pair) + this lane's 32-bit offset: the per-unit address update is two scalar adds
template <int S> __device__ __forceinline__ void dma_unit_s(const uint8_t *src, unsigned voff, unsigned lds_dst);
#define RWKV_DMA_UNIT_S(S_, BODY)                                                                                                      \
    template <> __device__ __forceinline__ void dma_unit_s<S_>(const uint8_t *src, unsigned voff, unsigned lds_dst)                    \
    {                                                                                                                                  \
        unsigned keep;                                                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" BODY "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds_dst) : "memory"); \
    }
#define RWKV_DMA_LS(OFF) "global_load_lds_dwordx4 %1, %2 offset:" #OFF " nt\n\t"
RWKV_DMA_UNIT_S(2, RWKV_DMA_LS(0) RWKV_DMA_LS(1024))
RWKV_DMA_UNIT_S(4, RWKV_DMA_LS(0) RWKV_DMA_LS(1024) RWKV_DMA_LS(2048) RWKV_DMA_LS(3072))
#undef RWKV_DMA_LS
#undef RWKV_DMA_UNIT_S
template <> __device__ __forceinline__ void dma_unit_s<5>(const uint8_t *src, unsigned voff, unsigned lds_dst)      // (the offset field ends at 4095)
{
    dma_unit_s<4>(src, voff, lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src + 4096), "s"(lds_dst + 4096u) : "memory");
}

// The loader wave of a tile-form kernel: units of S KiB (S k-blocks of one tile), in order.  Same rules as RingLoader (kernels.hip.h):
// one asm statement per unit, vmcnt READ instead of waited on, every wait bounded -- and, because one wave's instruction issue IS the
// stream's ceiling (the first version of this loader spent 50 instructions per unit and streamed at 23 KB/us per CU where the row
// form's reaches 26: profiles/r05/tile_check.txt), ALL state scalar: the unit's address is an SGPR pair advanced by two scalar adds,
// `landed` is stored by all lanes at once (lane 0 into the control block, the others into dump slots).  Consumers free units by
// per-wave counters (unit u belongs to wave u % NC, which takes its units in order), so the first unit still in use is
// min_w (NC * freed[w] + w): looked at only when the cached value says the ring is full.
// Who takes which unit when the stream has a HEAD (round 6; RUN = 1 only): the first HEAD units belong to the three consumer waves that do not stage
// (waves 4 .. 6, round robin), who copy them into REGISTERS as they land -- while waves 0 .. 3 are still staging -- and hand their ring slots back: the ring's
// capacity grows by HEAD units, the loader is not stopped by a slot whose owner is busy elsewhere (its tail only moves over LEADING free units).  Behind the
// head: round robin over all seven.  The k-th unit of wave w:
template <int HEAD> __host__ __device__ constexpr int tile_unit_of(int w, int k)
{
    constexpr int NE = HEAD / TILE_NSTASH;
    return w >= TILE_NWP ? (k < NE ? TILE_NSTASH * k + (w - TILE_NWP) : HEAD + NC * (k - NE) + w) : HEAD + NC * k + w;
}
template <int S, int RUN = 1, int HEAD = 0> struct TileLoader {
    TileCtl *tc;
    unsigned ring, nu;
    unsigned issued = 0, pub = 0, tailu = 0, pos = 0;
    unsigned voff;               // lane * 16
    unsigned landed_lds;         // LDS byte address this lane stores `landed` to: lane 0 the control block's word, lane l dump[l]
    int lane;
    bool dead = false;
    unsigned fail = 0;
    __device__ __forceinline__ TileLoader(TileCtl *tc_, unsigned ring_, int nu_, int lane_)
        : tc(tc_), ring((unsigned)__builtin_amdgcn_readfirstlane((int)ring_)), nu((unsigned)nu_), voff((unsigned)lane_ << 4),
          landed_lds(lane_ == 0 ? lds_addr(&tc_->landed) : lds_addr(&tc_->dump[lane_])), lane(lane_)
    {
        asm volatile("" : "+s"(ring));
    }
    __device__ __forceinline__ unsigned in_flight() const
    {
        const unsigned v = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 7);
        return (v & 0xfu) | ((v >> 18) & 0x30u);
    }
    // (asm: a store the compiler sees through a generic pointer becomes a FLAT store + s_waitcnt vmcnt(0) -- it would sit in this wave's
    // vmcnt, which IS the landing count, and drain the DMA queue at every announcement)
    __device__ __forceinline__ void store_landed(unsigned units) { asm volatile("ds_write_b32 %0, %1" ::"v"(landed_lds), "v"(units) : "memory"); }
    // (this wave issues no vector memory operation but its DMA pieces: issued * S >= in_flight, always)
    __device__ __forceinline__ void poll_landed()
    {
        const unsigned units = (issued * (unsigned)S - in_flight()) / (unsigned)S;
        if (units != pub) { pub = units; store_landed(units); }
    }
    __device__ __forceinline__ void refresh_tail()
    {
        const unsigned f = __hip_atomic_load(&tc->freed[lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        // the f-th unit wave w takes is unit ((f / RUN) NC + w) RUN + f % RUN: the first one it has NOT taken yet (HEAD > 0: tile_unit_of)
        const unsigned v = HEAD > 0 ? (unsigned)tile_unit_of<HEAD>(lane & 7, (int)f)
                                    : ((f / (unsigned)RUN) * (unsigned)NC + (unsigned)(lane & 7)) * (unsigned)RUN + f % (unsigned)RUN;
        unsigned m = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
#pragma unroll
        for (int w = 1; w < NC; w++) { const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)v, w); m = o < m ? o : m; }
        tailu = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }
    __device__ __forceinline__ bool room2() const { return issued + 2u - tailu <= nu; }
    // TWO units (2 S KiB, contiguous in the image and -- nu is even, pairs are aligned -- in the ring) per round of bookkeeping
    template <int DEPTH> __device__ __forceinline__ void pair(const uint8_t *src)
    {
        if (!room2()) {
            for (int it = 0; !dead; it++) {
                refresh_tail();
                if (room2()) break;
                poll_landed();
                if (it >= GLDS_SPIN) { dead = true; fail = 1u; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        wait_vm<(DEPTH > 2 * S ? DEPTH : 2 * S) - 2 * S>();
        const unsigned dst = ring + pos * (unsigned)(S * 1024);
        dma_unit_s<S>(src, voff, dst);
        dma_unit_s<S>(src + S * 1024, voff, dst + (unsigned)(S * 1024));
        pos = pos + 2 == nu ? 0u : pos + 2;
        issued += 2;
        poll_landed();
    }
    __device__ __forceinline__ void finish()
    {
        int it = 0;
        for (; it < GLDS_SPIN && in_flight() != 0u; it++) { poll_landed(); __builtin_amdgcn_s_sleep(1); }
        fail = it >= GLDS_SPIN ? 4u : fail;
        wait_vm<0>();
        pub = issued;
        store_landed(issued);
    }
};

// ---- shared by the tile-form kernels ----------------------------------------------------------------------------------------------
// plain-vector prologue (k_attout: the gated wkv vector; k_ffnv: the 4 D hidden units as ONE vector), run by waves 0..3: scale and
// offset from the producer's per-workgroup partials, then quantise and stage n elements (n % 1024 == 0 here).  NQ = quads per thread.
template <int NQ>
__device__ __forceinline__ void tile_vec(const float *vec, const double *partS, const float *partM, int n_part, int n, double *red, unsigned *xq,
                                         TileCtl *tc, unsigned long long *tl)
{
    constexpr int NTP = NT / 2, NWP = NTP / 64;
    const int lane = threadIdx.x & 63, wave = wave_id(), nqd = n >> 2;
    float *bc = reinterpret_cast<float *>(red + RED_BC);
    unsigned *spin = reinterpret_cast<unsigned *>(bc + 8);
    double ps = partS[(int)threadIdx.x < n_part ? threadIdx.x : 0];
    float pm = partM[(int)threadIdx.x < n_part ? threadIdx.x : 0];
    f32x4 vl[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NTP;
        vl[i] = reinterpret_cast<const f32x4 *>(vec)[qd < nqd ? qd : nqd - 1];
    }
    tl_stamp(tl, 1);
    if (threadIdx.x == 0) *spin = 0u;
    __syncthreads();   // order
    TL_BAR(tl, 2);
    if ((int)threadIdx.x >= n_part) { ps = 0.0; pm = 0.f; }
    float *redf = reinterpret_cast<float *>(red + RED_MAX);
    const double ws = wave_sum(ps);
    const float wm = wave_max(pm);
    if (lane == 0) { red[RED_OFFS + wave] = ws; redf[wave] = wm; }
    tl_stamp(tl, 3);
    if (lane == 0) __hip_atomic_fetch_add(spin, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned f_ = 0u;
    wait_count(spin, NWP, f_);
    double ts = 0.0; float tm = 0.f;
#pragma unroll
    for (int i = 0; i < NWP; i++) { ts += red[RED_OFFS + i]; tm = fmaxf(tm, redf[i]); }
    tl_stamp(tl, 4);
    unsigned long long sq = 0ull;
    const float inv = inv_scale(tm);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const int qd = threadIdx.x + i * NTP;
        if (qd < nqd) {
            const float v[4] = {vl[i][0], vl[i][1], vl[i][2], vl[i][3]};
            sq += stage_quad_t(xq, qd, v, inv);
        }
    }
    {
        const unsigned lo = wave_sum_dpp((unsigned)sq & 0xffffu), hi = wave_sum_dpp((unsigned)(sq >> 16));
        if (lane == 0) __hip_atomic_fetch_add(&tc->sq[0], (unsigned long long)lo + ((unsigned long long)hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (threadIdx.x == 0) { bc[0] = (float)ts; bc[4] = tm; }
    if (lane == 0) __hip_atomic_fetch_add(&tc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// waves TILE_NWP .. NC - 1 fetch the epilogues' inputs of the workgroup's rows into LDS while waves 0..3 stage: the wave's stores are announced on
// the count every consumer waits for before its first unit.  (a tile's finisher may be ANY consumer wave, also one that took no unit
// beside these waves -- D = 5120 / 2048 with their small tiles --, and the order barrier alone does not order these stores against it.)
__device__ __forceinline__ void tile_stash_done(TileCtl *tc, int lane)
{
    if (lane == 0) __hip_atomic_fetch_add(&tc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the loader wave's whole life: zero the control block and the tile sums, first units before the order barrier, the stream, the landing
template <int S, int UPT, int RUN, int TH, int PRE = RWKV_RING_PRE, int HEAD = 0, class Src>
__device__ __forceinline__ unsigned tile_loader(int NU, int ntile, unsigned char *ring, int ns, TileCtl *tc, int *tsum, int lane, Src unit_src,
                                                unsigned long long *tl = nullptr)
{
    static_assert(UPT % 2 == 0, "pairs of units never straddle a tile");
    loader_clean_slate();
    for (int i = lane; i < (int)(sizeof(TileCtl) / 4); i += 64) reinterpret_cast<unsigned *>(tc)[i] = 0u;
    for (int i = lane; i < ntile * TH * 3; i += 64) tsum[i] = 0;
    static_assert(HEAD == 0 || RUN == 1, "a head of the stream: round robin only");
    TileLoader<S, RUN, HEAD> ld(tc, lds_addr(ring), ns, lane);
    // units in stream order = tile after tile, S k-blocks at a time: contiguous within a tile; the loader moves PAIRS of units
    const uint8_t *src = unit_src(0);
    int u = 0, c = 0;
    auto next = [&]() {
        u += 2; c += 2; src += 2 * S * 1024;
        if (c == UPT) { c = 0; src = unit_src(u < NU ? u : 0); }
    };
    const int pre = PRE < ns - 2 ? PRE : ns - 2;
    for (; u < NU && u < pre; next()) ld.template pair<RWKV_RING_PRE_DEPTH>(src);
    TL_BAR(tl, 3);
    __syncthreads();   // order: the control block is zero, the prologue's requests are in the pipe
    TL_BAR(tl, 1);
#if RWKV_TL_STREAM == 1
    tl_stamp(tl, 1);
    for (; u < NU; next()) {
        ld.template pair<(RWKV_RING_DEPTH < 63 ? RWKV_RING_DEPTH : 63)>(src);
        if (u + 2 == (NU / 4 & ~1)) tl_stamp(tl, 3);
        if (u + 2 == (NU / 2 & ~1)) tl_stamp(tl, 4);
        if (u + 2 == (3 * NU / 4 & ~1)) tl_stamp(tl, 5);
    }
#endif
    // (TEST build -DRWKV_TEST_DROP_GROUP=1, tests/test_engine_gpu.py: the loader "loses" the workgroup's last pair of units -- the consumers'
    // bounded wait must give up and the call must fail with RWKV_E_DEVICE instead of returning garbage)
    for (; u < NU - (RWKV_TEST_DROP_GROUP ? 2 : 0); next()) ld.template pair<(RWKV_RING_DEPTH < 63 ? RWKV_RING_DEPTH : 63)>(src);
    ld.finish();
    return ld.fail;
}

// exact integer sum of tile t's row `row`: M = D0 + 2^8 D1 + 2^16 D2 over the three limb planes
template <int TH>
__device__ __forceinline__ long long tile_row_sum(const int *tsum, int t, int row)
{
    const int *p = tsum + (t * TH + row) * 3;
    return (long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
           + 256ll * (long long)__hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
           + 65536ll * (long long)__hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// per-vector constant of the signed-limb form: 128 * sum_j (q_j + 2^22) - 16384 * 65793 * N
__device__ __forceinline__ double tile_cA(const TileCtl *tc, int m, double n)
{
    return 128.0 * (double)__hip_atomic_load(&tc->sq[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + TILE_CN * n;
}

// per-wave state of a consumer's walk through the ring: units it has copied out (what it announces in freed[wave]), the last `landed` it saw
struct TileWalk { unsigned taken = 0, seen = 0; unsigned long long waited = 0; };
// wait for unit u, copy its S KiB out of ring slot p, hand the slot back
template <int S>
__device__ __forceinline__ void tile_take(const unsigned char *ring, unsigned p, int u, TileCtl *tc, int wave, int lane, TileWalk &wk, unsigned &fail, u32x4 (&w)[S])
{
    if ((int)(wk.seen - (unsigned)(u + 1)) < 0) {       // (`landed` only grows: what an earlier look saw still holds)
#if RWKV_TL_STREAM == 1
        const unsigned long long w0 = wall_clock64();
#endif
        bool ok = false;
        for (int it = 0; it < GLDS_SPIN; it++) {
            wk.seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tc->landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            if ((int)(wk.seen - (unsigned)(u + 1)) >= 0) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        fail = ok ? fail : 2u;
#if RWKV_TL_STREAM == 1
        wk.waited += wall_clock64() - w0;
#endif
    }
    const u32x4 *wp = reinterpret_cast<const u32x4 *>(ring + (size_t)p * (S * 1024)) + lane;
#pragma unroll
    for (int s = 0; s < S; s++) w[s] = wp[s * 64];
    wk.taken++;
    // (relaxed + a compiler barrier: a wave's LDS operations execute in order, so the loader that sees the count finds the reads done;
    // a release store would make the wave WAIT for its reads before it may even request the activation limbs)
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(&tc->freed[wave], wk.taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the three non-staging consumer waves copy their units of the stream's head into registers while the vectors are being staged (tile_unit_of)
template <int S, int HEAD>
__device__ __forceinline__ void tile_pretake(int NU, const unsigned char *ring, int ns, TileCtl *tc, int wave, int lane, TileWalk &wk, unsigned &fail,
                                             u32x4 (&wpre)[HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1][S])
{
    if constexpr (HEAD > 0) {
        if (HEAD > ns) fail = 2u;                        // (the head lies in the ring's first lap: slot = unit; engine.hip sizes the ring)
#pragma unroll
        for (int k = 0; k < HEAD / TILE_NSTASH; k++) {
            const int u = TILE_NSTASH * k + (wave - TILE_NWP);
            if (u < NU && HEAD <= ns) tile_take<S>(ring, (unsigned)u, u, tc, wave, lane, wk, fail, wpre[k]);
        }
    }
}

// The consumer waves' streaming loop (wave < NC).  A tile is TH rows (16: the chunk path's image; 4: the decode-only image of widths
// whose channels do not split into 16-row blocks per workgroup) x K inputs; a fragment (1 KiB, one wave instruction) holds TH rows x
// 64 / TH 16-byte pieces of k: lane l = piece l / TH of row l % TH.  Unit u (tile u / UPT, fragments S (u % UPT) ..) belongs to the wave
// that owns its RUN of units (run u / RUN -> wave (u / RUN) % NC; RUN = 1: round robin; RUN = UPT: a wave multiplies whole small tiles;
// HEAD > 0: tile_unit_of): wait for it, copy its S KiB out of the ring, hand it back (tile_take), multiply it with the limbs of vector
// vec_of(tile); when the wave's last unit of a tile is in, its partial sums join the tile's in LDS and the unit count; the wave that
// completes the count calls on_tile(t) (all lanes; the tile's row sums are final: tile_row_sum).
template <int TH, int S, int UPT, int RUN, int HEAD, class VecOf, class OnTile>
__device__ __forceinline__ void tile_consume(int NU, const unsigned char *ring, int ns, TileCtl *tc, int *tsum, const unsigned *xq, int xvd_t,
                                             int wave, int lane, unsigned &fail, TileWalk &wk, const u32x4 (&wpre)[HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1][S],
                                             VecOf vec_of, OnTile on_tile, unsigned long long *tl = nullptr)
{
    static_assert(UPT % RUN == 0, "a run of units never straddles a tile");
    static_assert(HEAD == 0 || (RUN == 1 && HEAD % TILE_NSTASH == 0), "a head of the stream: round robin only, whole rounds of the three waves");
    constexpr int PPB = 64 / TH;                         // 16-byte pieces of k per row and fragment
    const int pc = lane / TH, r = lane % TH;
    int acc0 = 0, acc1 = 0, acc2 = 0, cnt = 0;
    tile_i32x4 ma = tile_i32x4{0, 0, 0, 0}, mb = tile_i32x4{0, 0, 0, 0};
    // multiply unit u (in registers) and, when this wave's next unit `un` lies in another tile (or nowhere), add the tile's partial sums up
    auto mul = [&](const u32x4 (&w)[S], int u, int un) {
        const int t = u / UPT, c = u - t * UPT;
        if constexpr (TILE_MFMA && TH == 16) {
            // a 16-row fragment IS the B operand of v_mfma_i32_16x16x64_i8 and the staged vector's (k-block, piece, limb) order its A operand with
            // the three limbs as rows 0..2: one 16-byte LDS read and one matrix instruction per fragment where the VALU form needs three reads and
            // twelve dot instructions.  Rows 3..15 of A are whatever those lanes read (limb 2 again): their outputs are never looked at.
            // The sums are the same exact integers.
            const u32x4 *xp = reinterpret_cast<const u32x4 *>(xq + vec_of(t) * xvd_t) + ((c * S) * PPB + pc) * 3 + (r < 3 ? r : 2);
#pragma unroll
            for (int s = 0; s < S; s++) {
                const u32x4 x = xp[s * PPB * 3];
                const tile_i32x4 af = tile_i32x4{(int)x[0], (int)x[1], (int)x[2], (int)x[3]}, bf = tile_i32x4{(int)w[s][0], (int)w[s][1], (int)w[s][2], (int)w[s][3]};
                if (s & 1) mb = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, mb, 0, 0, 0);
                else ma = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, ma, 0, 0, 0);
            }
        } else if constexpr (TILE_MFMA4 && TH == 4) {
            // a 4-row fragment (lane l = piece l / 4 of row l % 4; 16 pieces = 256 inputs) read as a B operand is 16 VIRTUAL rows n = (row, piece % 4)
            // whose k-piece j is the row's piece 4 j + (piece % 4).  A's rows are (limb b, q): A[(b, q)][j] = limb b of piece 4 j + q, so that
            // D[(b, q)][(row, q)] -- the diagonal blocks -- are the products that belong together; a row's limb sum is the sum of its four q.
            // Twelve of A's sixteen rows carry data; one instruction per fragment, like the 16-row form.
            const int m = (lane & 15) < 12 ? (lane & 15) : 11;
            const u32x4 *xp = reinterpret_cast<const u32x4 *>(xq + vec_of(t) * xvd_t) + ((c * S) * PPB + 4 * (lane >> 4) + (m & 3)) * 3 + (m >> 2);
#pragma unroll
            for (int s = 0; s < S; s++) {
                const u32x4 x = xp[s * PPB * 3];
                const tile_i32x4 af = tile_i32x4{(int)x[0], (int)x[1], (int)x[2], (int)x[3]}, bf = tile_i32x4{(int)w[s][0], (int)w[s][1], (int)w[s][2], (int)w[s][3]};
                if (s & 1) mb = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, mb, 0, 0, 0);
                else ma = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, ma, 0, 0, 0);
            }
        } else {
            const u32x4 *xp = reinterpret_cast<const u32x4 *>(xq + vec_of(t) * xvd_t) + ((c * S) * PPB + pc) * 3;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const u32x4 x0 = xp[s * PPB * 3], x1 = xp[s * PPB * 3 + 1], x2 = xp[s * PPB * 3 + 2];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    acc0 = __builtin_amdgcn_sdot4((int)w[s][d], (int)x0[d], acc0, false);
                    acc1 = __builtin_amdgcn_sdot4((int)w[s][d], (int)x1[d], acc1, false);
                    acc2 = __builtin_amdgcn_sdot4((int)w[s][d], (int)x2[d], acc2, false);
                }
            }
        }
        cnt++;
        if (un >= NU || un / UPT != t) {
            if constexpr (TILE_MFMA4 && TH == 4) {
                // accumulator image: lane (n = l % 16, g = l / 16) holds D[(g, i)][n] in register i: limb g of virtual row n = (row n % 4, q = n / 4)
                // is register n / 4; the four q of a row sit 4 lanes apart inside the 16 lanes
                const int n = lane & 15;
                const tile_i32x4 sa = ma + mb;
                int v = n < 4 ? sa[0] : n < 8 ? sa[1] : n < 12 ? sa[2] : sa[3];
                v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
                if (n < 4 && lane < 48) __hip_atomic_fetch_add(tsum + (t * TH + n) * 3 + (lane >> 4), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ma = mb = tile_i32x4{0, 0, 0, 0};
            } else if (TH == 4) {
                // 16 lanes hold pieces of one row: fold the pieces 4 apart inside every row of 16 lanes first (DPP), then the four rows of
                // 16 lanes meet in LDS like the four k-quarters of the 16-row form
                acc0 += __builtin_amdgcn_update_dpp(0, acc0, 0x124, 0xf, 0xf, true); acc0 += __builtin_amdgcn_update_dpp(0, acc0, 0x128, 0xf, 0xf, true);
                acc1 += __builtin_amdgcn_update_dpp(0, acc1, 0x124, 0xf, 0xf, true); acc1 += __builtin_amdgcn_update_dpp(0, acc1, 0x128, 0xf, 0xf, true);
                acc2 += __builtin_amdgcn_update_dpp(0, acc2, 0x124, 0xf, 0xf, true); acc2 += __builtin_amdgcn_update_dpp(0, acc2, 0x128, 0xf, 0xf, true);
            }
            if constexpr (TILE_MFMA && TH == 16) {      // (accumulator image: lane n < 16 holds row n's three limb sums in registers 0..2)
                if (lane < 16) {
                    int *ts = tsum + (t * TH + lane) * 3;
                    __hip_atomic_fetch_add(ts + 0, ma[0] + mb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(ts + 1, ma[1] + mb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(ts + 2, ma[2] + mb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ma = mb = tile_i32x4{0, 0, 0, 0};
            } else if constexpr (TILE_MFMA4 && TH == 4) {
            } else if (TH == 16 || (lane & 15) < 4) {
                int *ts = tsum + (t * TH + r) * 3;
                __hip_atomic_fetch_add(ts + 0, acc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ts + 1, acc1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ts + 2, acc2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            unsigned old = 0u;
            if (lane == 0) old = __hip_atomic_fetch_add(&tc->tcnt[t], (unsigned)cnt, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
            if (old + (unsigned)cnt == (unsigned)UPT) on_tile(t);
            acc0 = acc1 = acc2 = 0; cnt = 0;
        }
    };
    if constexpr (HEAD > 0) {
        constexpr int NE = HEAD / TILE_NSTASH;
        if (wave >= TILE_NWP) {
#pragma unroll
            for (int k = 0; k < NE; k++) {
                const int u = tile_unit_of<HEAD>(wave, k);
                if (u < NU) mul(wpre[k], u, tile_unit_of<HEAD>(wave, k + 1));
            }
        }
        unsigned p = (unsigned)(HEAD + wave) % (unsigned)ns;
#pragma unroll 1
        for (int u = HEAD + wave; u < NU; u += NC) {
            u32x4 w[S];
            tile_take<S>(ring, p, u, tc, wave, lane, wk, fail, w);
            p += (unsigned)NC;
            while (p >= (unsigned)ns) p -= (unsigned)ns;
            mul(w, u, u + NC);
        }
    } else {
        unsigned p = (unsigned)(wave * RUN) % (unsigned)ns;      // ring position of the wave's next unit: advanced unit by unit, run by run
        for (int run = wave; run * RUN < NU; run += NC) {
#pragma unroll 1
            for (int cc = 0; cc < RUN; cc++) {
                const int u = run * RUN + cc;
                u32x4 w[S];
                tile_take<S>(ring, p, u, tc, wave, lane, wk, fail, w);
                p = p + 1 == (unsigned)ns ? 0u : p + 1;
                mul(w, u, cc + 1 < RUN ? u + 1 : (run + NC) * RUN);
            }
            p += (unsigned)((NC - 1) * RUN);
            while (p >= (unsigned)ns) p -= (unsigned)ns;
        }
    }
#if RWKV_TL_STREAM == 1
    if (tl && lane == 0) tl[((size_t)blockIdx.x * NW + wave) * 8 + 2] = wk.waited;
#endif
}

// Units (of S KiB) a tile-form loader requests IN FRONT of the workgroup's order barrier, per kernel class.  The row-form default (RWKV_RING_PRE = 8
// rows) puts 32 KiB of weights into the CU's in-order memory path ahead of the prologue's loads (rule 1): round 6 measured the
// kernels whose prologue is their critical path with fewer (profiles/r06/tile_pre_ab.txt; 4-row tiles at D = 5120: tile_pre_14b_ab.txt).
#ifndef RWKV_TILE_PRE_ATT         // (0 without the head in registers; 4 with it)
#define RWKV_TILE_PRE_ATT 4
#endif
#ifndef RWKV_TILE_PRE_ATTOUT
#define RWKV_TILE_PRE_ATTOUT 4
#endif
#ifndef RWKV_TILE_PRE_FRK
#define RWKV_TILE_PRE_FRK 4
#endif
#ifndef RWKV_TILE_PRE_FV
#define RWKV_TILE_PRE_FV RWKV_RING_PRE
#endif
// Units of the stream's HEAD that the three non-staging consumer waves keep in registers (tile_unit_of; 16-row tiles, multiples of 3; 0: none).
// Measured at 7B (profiles/r06/tile_head_ab.txt): K/V/R with 9 units in registers and 4 in front of the barrier 12.3 -> 12.0 us (+0.5-0.7 % on the
// token); the other classes +-0, 4-row tiles (14B) -1 ... 0 %; streaming 16-24 units ahead of the prologue loses whatever the capacity (its loads
// return behind the stream's).
#ifndef RWKV_TILE_HEAD_ATT
#define RWKV_TILE_HEAD_ATT 9
#endif
#ifndef RWKV_TILE_PRE4_ATT        // K/V/R on 4-row tiles (no head there)
#define RWKV_TILE_PRE4_ATT 0
#endif
#ifndef RWKV_TILE_HEAD_ATTOUT
#define RWKV_TILE_HEAD_ATTOUT 0
#endif
#ifndef RWKV_TILE_HEAD_FRK
#define RWKV_TILE_HEAD_FRK 0
#endif
#ifndef RWKV_TILE_HEAD_FV
#define RWKV_TILE_HEAD_FV 0
#endif
#ifndef RWKV_TILE_PRE_FV4        // k_ffnv_t on 4-row tiles (D = 5120: 80 KB of hidden units to stage per workgroup): 14B 350.3 -> 354.9 tokens/s with none in front
#define RWKV_TILE_PRE_FV4 0
#endif
// Every tile-form kernel: workgroup b owns CPW = TH * TPC consecutive channels -- TPC tiles of TH rows per row class (TH = 16, TPC = 1: D = 4096
// on 256 CUs, the chunk path's own image; TH = 4: a decode-only image, TPC = 5 at D = 5120, 2 at D = 2048).  KBT = fragments (1 KiB) of a
// tile along K (K TH / 1024), S = fragments (KiB) per ring unit, SD = ceil(D / 1024).  The argument blocks are the row-form kernels' (site,
// epilogue inputs and outputs, D, ns = ring units, tl, herr; w unused) + the image.
struct TileImage {
    const uint8_t *bimg;          // tile image of this layer's matrix (k_bimage: tile id = class * CB + block, TH rows per tile, signed bytes)
    int CB;                       // TH-row blocks per row class (channels / TH)
};
struct FfnRKTArgs { FfnRKArgs a; TileImage im; };
struct AttTArgs { AttArgs a; TileImage im; };
struct AttOutTArgs { AttOutArgs a; TileImage im; };
struct FfnVTArgs { FfnVArgs a; TileImage im; };
// units a wave takes in a row.  1 = round robin: even shares whatever the tile count (14B: run 1 / 2 / 4 = 337 / 334 / 320 tokens/s, 15 tiles of
// K/V/R over 7 waves in whole tiles is 3 : 2); at 1B5 (two tiles of 8 KiB per class) whole tiles are the better of three slower-than-row-form
// choices (profiles/r05/tile_run_ab.txt)
constexpr int tile_run(int TH, int UPT, int TPC) { return TH == 16 || TPC > 2 ? 1 : (UPT < 4 ? UPT : 4); }
// LDS in front of the ring (bytes), per kernel: must match the carving at the head of each kernel (CPW = channels per workgroup)
constexpr size_t tile_fixed_frk(int D, int CPW) { return RED_BYTES + (size_t)2 * (D / 16) * 48 + (size_t)13 * CPW * 4 + (size_t)5 * CPW * 12 + sizeof(TileCtl); }
constexpr size_t tile_fixed_att(int D, int CPW) { return RED_BYTES + (size_t)3 * (D / 16) * 48 + (size_t)3 * CPW * 4 + (size_t)4 * CPW * 8 + (size_t)2 * CPW * 4 + (size_t)2 * CPW * 8 + (size_t)2 * CPW * 4 + (size_t)3 * CPW * 12 + sizeof(TileCtl); }
constexpr size_t tile_fixed_attout(int D, int CPW) { return RED_BYTES + (size_t)(D / 16) * 48 + (size_t)4 * CPW * 8 + (size_t)CPW * 12 * 4 + (size_t)CPW * 4 + (size_t)CPW * 10 * 8 + (size_t)CPW * 12 + sizeof(TileCtl); }
constexpr size_t tile_fixed_fv(int D, int CPW) { return RED_BYTES + (size_t)(4 * D / 16) * 48 + (size_t)4 * CPW * 8 + (size_t)CPW * 16 * 4 + (size_t)CPW * 4 + (size_t)CPW * 4 + (size_t)CPW * 10 * 8 + (size_t)CPW * 12 + sizeof(TileCtl); }

// ln2 site -> ffn_r GEMV + sigmoid, ffn_k GEMV + relu^2 (rwkv.cu:557-573): 5 row classes (0..3 = ffn_k outputs 4 i + q, 4 = ffn_r output i) x TPC tiles
template <int SD, int S, int KBT, int TH, int TPC>
__global__ __launch_bounds__(NT) void k_ffn_rk_t(FfnRKTArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FfnRKArgs &a = ta.a;
    constexpr int CPW = TH * TPC, NTILE = 5 * TPC, UPT = KBT / S, NU = NTILE * UPT, RUN = tile_run(TH, UPT, TPC);
    constexpr int HEAD = TH == 16 && RUN == 1 ? RWKV_TILE_HEAD_FRK : 0, NE = HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1;
    static_assert(HEAD <= NU && HEAD % TILE_NSTASH == 0, "the head: whole rounds of the three non-staging waves, inside the stream");
    static_assert(KBT % S == 0 && NTILE <= 32, "whole units per tile; TileCtl::tcnt");
    double *red = reinterpret_cast<double *>(smem);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int xvd_t = (D >> 4) * 12;                       // dwords of one staged vector: [16-byte piece][limb][4]
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    // LDS: [scratch][2 staged vectors][stash: rs[5 CPW], r_fv[4 CPW], o_fv[4 CPW]][tile sums][TileCtl][ring]
    unsigned *stash = xq + 2 * xvd_t;
    int *tsum = reinterpret_cast<int *>(stash + 13 * CPW);
    TileCtl *tc = reinterpret_cast<TileCtl *>(tsum + NTILE * TH * 3);
    unsigned char *ring = reinterpret_cast<unsigned char *>(tc + 1);
    RWKV_ARGS_NOW(a.x, a.st.C, a.st.TC, a.st.maxC, a.dy.B, a.dy.pd, a.dy.pf, a.dy.n_part);
    const int cb0 = blockIdx.x * TPC, ch0 = blockIdx.x * CPW;
    tl_stamp(a.tl, 0);
    auto unit_src = [&](int u) {          // tile t = class t / TPC, block cb0 + t % TPC
        const int t = u / UPT, c = u - t * UPT;
        return ta.im.bimg + ((size_t)((t / TPC) * ta.im.CB + cb0 + t % TPC) * KBT + (size_t)c * S) * 1024;
    };
    double part = 0.0;
    float pmax = 0.f;
    unsigned fail = 0u;
    if (wave == NC) {
        fail = tile_loader<S, UPT, RUN, TH, RWKV_TILE_PRE_FRK, HEAD>(NU, NTILE, ring, a.ns, tc, tsum, lane, unit_src, a.tl);
        tl_stamp(a.tl, 2);
    } else {
        if (wave < TILE_NWP) {
            tile_site<2, SD>(a.st, a.dy, a.x, D, red, xq, xvd_t, true, tc, a.tl);
        } else {
            // waves 4..6 have nothing to do until the vectors are staged: they fetch the epilogues' inputs of the workgroup's 5 CPW rows into
            // LDS (row sums of the unsigned weights, ffn_v's scale and offset of the 4 CPW hidden units): requested in front of the order
            // barrier, stored behind it (a wave that waited for its loads in front of the barrier would hold the prologue waves there)
            const int n = wave == 4 ? 5 * CPW : 4 * CPW;
            unsigned v0 = 0u, v1 = 0u;
            const unsigned *src = wave == 4 ? a.rs + ch0 * 5 : reinterpret_cast<const unsigned *>(wave == 5 ? a.r_fv + ch0 * 4 : a.o_fv + ch0 * 4);
            if (lane < n) v0 = src[lane];
            if (lane + 64 < n) v1 = src[lane + 64];
            TL_BAR(a.tl, 1);
            __syncthreads();   // order
            TL_BAR(a.tl, 2);
            unsigned *dst = stash + (wave == 4 ? 0 : wave == 5 ? 5 * CPW : 9 * CPW);
            if (lane < n) dst[lane] = v0;
            if (lane + 64 < n) dst[lane + 64] = v1;
            tile_stash_done(tc, lane);
        }
        float *bc = reinterpret_cast<float *>(red + RED_BC);
        TileWalk wk;
        u32x4 wpre[NE][S];
        if (wave >= TILE_NWP) tile_pretake<S, HEAD>(NU, ring, a.ns, tc, wave, lane, wk, fail, wpre);
        wait_count(&tc->staged, TILE_NWP + TILE_NSTASH, fail);
        const double sck = scale_of(bc[4]), scr = scale_of(bc[5]);
        const float Sk = bc[0], Sr = bc[1];
        const double cAk = tile_cA(tc, 0, (double)D), cAr = tile_cA(tc, 1, (double)D);
        tl_stamp(a.tl, 5);
        auto on_tile = [&](int t) {          // tile t = class q, block sub: this lane's row is channel ch0 + sub TH + lane
            if (lane < TH) {
                const int q = t / TPC, li = (t % TPC) * TH + lane;
                const long long M = tile_row_sum<TH>(tsum, t, lane);
                const unsigned rs = stash[li * 5 + q];
                if (q < 4) {
                    const float val = (float)(sck * ((double)M + cAk + TILE_CU * (double)rs)) + Sk;
                    float h = val * (float)(val > 0.f);   // rwkv.cu:189-190
                    h = h * h;
                    const int hi = li * 4 + q;
                    const float hs = h * __uint_as_float(stash[5 * CPW + hi]);
                    a.hbuf[4 * ch0 + hi] = hs;
                    part += (double)(h * __uint_as_float(stash[9 * CPW + hi]));
                    pmax = fmaxf(pmax, fabsf(hs));
                } else {
                    const float val = (float)(scr * ((double)M + cAr + TILE_CU * (double)rs)) + Sr;
                    a.rgate[ch0 + li] = (float)(1.0 / (1.0 + exp(-(double)val)));   // rwkv.cu:212
                }
            }
        };
        tile_consume<TH, S, UPT, RUN, HEAD>(NU, ring, a.ns, tc, tsum, xq, xvd_t, wave, lane, fail, wk, wpre, [](int t) { return t / TPC < 4 ? 0 : 1; }, on_tile, a.tl);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();   // every wave is past its last read of the reduction scratch
    block_sum_max(part, pmax, red + RED_PART);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part; a.partM[blockIdx.x] = pmax; }
    ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ln1 site -> K, V, R dequant-GEMV -> WKV (rwkv.cu:535-545; kernels :351-392, :58-100, :221-259): 3 row classes x TPC tiles.  The tile
// finishers leave k's two exponentials, v and r in LDS; whoever completes a block's third tile runs its TH recurrences side by side.
template <int SD, int S, int KBT, int TH, int TPC>
__global__ __launch_bounds__(NT) void k_att_t(AttTArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AttArgs &a = ta.a;
    constexpr int CPW = TH * TPC, NTILE = 3 * TPC, UPT = KBT / S, NU = NTILE * UPT, RUN = tile_run(TH, UPT, TPC);
    constexpr int HEAD = TH == 16 && RUN == 1 ? RWKV_TILE_HEAD_ATT : 0, NE = HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1;
    static_assert(HEAD <= NU && HEAD % TILE_NSTASH == 0, "the head: whole rounds of the three non-staging waves, inside the stream");
    static_assert(KBT % S == 0 && NTILE <= 32 && TPC <= 8, "whole units per tile; TileCtl::tcnt / done");
    double *red = reinterpret_cast<double *>(smem);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int xvd_t = (D >> 4) * 12;
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    // LDS: [scratch][3 staged vectors][stash: rs[3 CPW] | aa, bb, uw, ew [4][CPW] f64 | ra, oa [2][CPW] f32 | e1, ek [2][CPW] f64 | v, r [2][CPW] f32][tile sums][TileCtl][ring]
    unsigned *stash = xq + 3 * xvd_t;
    double *sd = reinterpret_cast<double *>(stash + 3 * CPW);       // aa, bb, uw, ew    (3 CPW words: CPW is a multiple of 4, 8-byte aligned)
    float *sf = reinterpret_cast<float *>(sd + 4 * CPW);            // ra, oa
    double *ed = reinterpret_cast<double *>(sf + 2 * CPW);          // e1, ek
    float *ef = reinterpret_cast<float *>(ed + 2 * CPW);            // v, r
    int *tsum = reinterpret_cast<int *>(ef + 2 * CPW);
    TileCtl *tc = reinterpret_cast<TileCtl *>(tsum + NTILE * TH * 3);
    unsigned char *ring = reinterpret_cast<unsigned char *>(tc + 1);
    RWKV_ARGS_NOW(a.x, a.st.C, a.st.TC, a.st.maxC, a.dy.B, a.dy.pd, a.dy.pf, a.dy.n_part);
    const int cb0 = blockIdx.x * TPC, ch0 = blockIdx.x * CPW;
    tl_stamp(a.tl, 0);
    auto unit_src = [&](int u) {
        const int t = u / UPT, c = u - t * UPT;
        return ta.im.bimg + ((size_t)((t / TPC) * ta.im.CB + cb0 + t % TPC) * KBT + (size_t)c * S) * 1024;
    };
    double part = 0.0;
    float pmax = 0.f;
    unsigned fail = 0u;
    if (wave == NC) {
        fail = tile_loader<S, UPT, RUN, TH, (TH == 16 ? RWKV_TILE_PRE_ATT : RWKV_TILE_PRE4_ATT), HEAD>(NU, NTILE, ring, a.ns, tc, tsum, lane, unit_src, a.tl);
        tl_stamp(a.tl, 2);
    } else {
        const size_t so = (size_t)a.ctl->slot * a.slot_stride;
        if (wave < TILE_NWP) {
            tile_site<3, SD>(a.st, a.dy, a.x, D, red, xq, xvd_t, true, tc, a.tl);
        } else {
            // epilogue inputs of the CPW channels: wave 4 the 3 CPW row sums, wave 5 the state (aa | bb) and the decay terms (uw | ew), wave 6 att_out's scale / offset
            unsigned v0 = 0u, v1 = 0u;
            double d0 = 0.0, d1 = 0.0;
            auto dsrc = [&](int i) { const int g = i / CPW, j = i - g * CPW; return g == 0 ? a.saa[so + ch0 + j] : g == 1 ? a.sbb[so + ch0 + j] : g == 2 ? a.uw[ch0 + j] : a.ew[ch0 + j]; };
            if (wave == 4) { if (lane < 3 * CPW) v0 = a.rs[ch0 * 3 + lane]; }
            else if (wave == 5) { if (lane < 4 * CPW) d0 = dsrc(lane); if (lane + 64 < 4 * CPW) d1 = dsrc(lane + 64); }
            else if (lane < 2 * CPW) v0 = __float_as_uint(lane < CPW ? a.r_att[ch0 + lane] : a.o_att[ch0 + lane - CPW]);
            (void)v1;
            TL_BAR(a.tl, 1);
            __syncthreads();   // order
            TL_BAR(a.tl, 2);
            if (wave == 4) { if (lane < 3 * CPW) stash[lane] = v0; }
            else if (wave == 5) { if (lane < 4 * CPW) sd[lane] = d0; if (lane + 64 < 4 * CPW) sd[lane + 64] = d1; }
            else if (lane < 2 * CPW) sf[lane] = __uint_as_float(v0);
            tile_stash_done(tc, lane);
        }
        float *bc = reinterpret_cast<float *>(red + RED_BC);
        TileWalk wk;
        u32x4 wpre[NE][S];
        if (wave >= TILE_NWP) tile_pretake<S, HEAD>(NU, ring, a.ns, tc, wave, lane, wk, fail, wpre);
        wait_count(&tc->staged, TILE_NWP + TILE_NSTASH, fail);
        tl_stamp(a.tl, 5);
        auto on_tile = [&](int t) {
            const int q = t / TPC, sub = t % TPC;
            if (lane < TH) {
                const int li = sub * TH + lane;
                const long long M = tile_row_sum<TH>(tsum, t, lane);
                // (the class's scale, offset scalar and cA constant come out of LDS with the tile's sums, in the same batch of reads: a tile
                // completes 3 TPC times per workgroup, and nine more live registers through the streaming loop made this kernel spill)
                const double scq = scale_of(bc[4 + q]), cAq = tile_cA(tc, q, (double)D);
                const float val = (float)(scq * ((double)M + cAq + TILE_CU * (double)stash[li * 3 + q])) + bc[q];
                if (q == 0) { ed[li] = exp(sd[2 * CPW + li] + (double)val); ed[CPW + li] = exp((double)val); }      // exp(u + w + k), exp(k)
                else ef[(q - 1) * CPW + li] = val;
            }
            unsigned old = 0u;
            if (lane == 0) old = __hip_atomic_fetch_add(&tc->done[sub], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
            if (old == 2u && lane < TH) {            // the block's three tiles are in: the WKV recurrence and the gate of its TH channels (rwkv.cu:242-255)
                const int li = sub * TH + lane, g = ch0 + li;
                const double aa = sd[li], bb = sd[CPW + li], ew = sd[3 * CPW + li];
                const double e1 = *reinterpret_cast<volatile double *>(ed + li), ek = *reinterpret_cast<volatile double *>(ed + CPW + li);
                const float v = *reinterpret_cast<volatile float *>(ef + li), r = *reinterpret_cast<volatile float *>(ef + CPW + li);
                const double vv = (double)v;
                double y = (aa + e1 * vv) / (bb + e1);
                y = (1.0 / (1.0 + (double)expf(-r))) * y;       // rwkv.cu:250: exp of a float argument
                a.saa[so + g] = (aa + ek * vv) * ew;
                a.sbb[so + g] = (bb + ek) * ew;
                const float yf = (float)y;                       // att_out GEMV casts its input to f32 (rwkv.cu:290)
                const float ys = yf * sf[li];
                a.ybuf[g] = ys;
                part += (double)(yf * sf[CPW + li]);
                pmax = fmaxf(pmax, fabsf(ys));
            }
        };
        tile_consume<TH, S, UPT, RUN, HEAD>(NU, ring, a.ns, tc, tsum, xq, xvd_t, wave, lane, fail, wk, wpre, [](int t) { return t / TPC; }, on_tile, a.tl);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();
    block_sum_max(part, pmax, red + RED_PART);
    if (threadIdx.x == 0) { a.partS[blockIdx.x] = part; a.partM[blockIdx.x] = pmax; }
    ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// the row owners' partial tuple of a tile-form kernel: the finishers' lanes leave the accumulators of their rows in LDS (entry = the
// row's index in the workgroup), thread k < 12 adds the CPW entries up (kernels.hip.h site_publish does it for R lanes of every wave)
template <int NV>
__device__ __forceinline__ void tile_site_leave(const SiteAcc<NV> &acc, double *scr, int cpw, int li)
{
    float *scf = reinterpret_cast<float *>(scr + cpw * 8);
#pragma unroll
    for (int k = 0; k < 8; k++) scr[li * 8 + k] = acc.d[k];
#pragma unroll
    for (int k = 0; k < 4; k++) scf[li * 4 + k] = acc.f[k];
}
__device__ __forceinline__ void tile_site_publish(const SiteDyn &dy, const double *scr, int cpw)
{
    const float *scf = reinterpret_cast<const float *>(scr + cpw * 8);
    if (threadIdx.x < 8) {
        double t = 0.0;
        for (int i = 0; i < cpw; i++) t += scr[i * 8 + threadIdx.x];
        dy.pd[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
    } else if (threadIdx.x < 12) {
        float t = 0.f;
        for (int i = 0; i < cpw; i++) t = fmaxf(t, scf[i * 4 + (threadIdx.x - 8)]);
        dy.pf[(size_t)blockIdx.x * 4 + (threadIdx.x - 8)] = t;
    }
}

// att_out dequant-GEMV + residual through f32 (rwkv.cu:548-553); commits state xy; opens the ln2 site for its CPW rows: TPC tiles
template <int SD, int S, int KBT, int TH, int TPC>
__global__ __launch_bounds__(NT) void k_attout_t(AttOutTArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AttOutArgs &a = ta.a;
    constexpr int CPW = TH * TPC, NTILE = TPC, UPT = KBT / S, NU = NTILE * UPT, RUN = tile_run(TH, UPT, TPC);
    constexpr int HEAD = TH == 16 && RUN == 1 ? RWKV_TILE_HEAD_ATTOUT : 0, NE = HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1;
    static_assert(HEAD <= NU && HEAD % TILE_NSTASH == 0, "the head: whole rounds of the three non-staging waves, inside the stream");
    double *red = reinterpret_cast<double *>(smem);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int xvd_t = (D >> 4) * 12;
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    // LDS: [scratch][1 staged vector][stash: xold, lw, lb, prev2 [4][CPW] f64 | P [CPW][12] f32 | rs[CPW]][publish scratch CPW x 10 f64][tile sums][TileCtl][ring]
    double *sd = reinterpret_cast<double *>(xq + xvd_t);
    float *sp = reinterpret_cast<float *>(sd + 4 * CPW);
    unsigned *srs = reinterpret_cast<unsigned *>(sp + CPW * 12);
    double *scr = reinterpret_cast<double *>(srs + CPW);
    int *tsum = reinterpret_cast<int *>(scr + CPW * 10);
    TileCtl *tc = reinterpret_cast<TileCtl *>(tsum + NTILE * TH * 3);
    unsigned char *ring = reinterpret_cast<unsigned char *>(tc + 1);
    RWKV_ARGS_NOW(a.ybuf, a.partS, a.partM, a.n_part);
    const int cb0 = blockIdx.x * TPC, ch0 = blockIdx.x * CPW;
    tl_stamp(a.tl, 0);
    auto unit_src = [&](int u) {
        const int t = u / UPT, c = u - t * UPT;
        return ta.im.bimg + ((size_t)(cb0 + t) * KBT + (size_t)c * S) * 1024;
    };
    unsigned fail = 0u;
    if (wave == NC) {
        fail = tile_loader<S, UPT, RUN, TH, RWKV_TILE_PRE_ATTOUT, HEAD>(NU, NTILE, ring, a.ns, tc, tsum, lane, unit_src, a.tl);
        tl_stamp(a.tl, 2);
    } else {
        const double mean1 = a.lnstat[0], rstd1 = a.lnstat[1];
        const size_t so = (size_t)a.ctl->slot * a.slot_stride;
        if (wave < TILE_NWP) {
            tile_vec<SD>(a.ybuf, a.partS, a.partM, a.n_part, D, red, xq, tc, a.tl);
        } else {
            auto dsrc = [&](int i) { const int g = i / CPW, j = i - g * CPW; return g == 0 ? a.x[ch0 + j] : g == 1 ? a.lnw[ch0 + j] : g == 2 ? a.lnb[ch0 + j] : a.sdd[so + ch0 + j]; };
            double d0 = 0.0, d1 = 0.0;
            f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned v0 = 0u;
            if (wave == 4) { if (lane < 4 * CPW) d0 = dsrc(lane); if (lane + 64 < 4 * CPW) d1 = dsrc(lane + 64); }
            else if (wave == 5) { if (lane < 3 * CPW) p0 = reinterpret_cast<const f32x4 *>(a.st.P + (size_t)ch0 * 12)[lane]; }
            else if (lane < CPW) v0 = a.rs[ch0 + lane];
            TL_BAR(a.tl, 1);
            __syncthreads();   // order
            TL_BAR(a.tl, 2);
            if (wave == 4) { if (lane < 4 * CPW) sd[lane] = d0; if (lane + 64 < 4 * CPW) sd[lane + 64] = d1; }
            else if (wave == 5) { if (lane < 3 * CPW) reinterpret_cast<f32x4 *>(sp)[lane] = p0; }
            else if (lane < CPW) srs[lane] = v0;
            tile_stash_done(tc, lane);
        }
        float *bc = reinterpret_cast<float *>(red + RED_BC);
        TileWalk wk;
        u32x4 wpre[NE][S];
        if (wave >= TILE_NWP) tile_pretake<S, HEAD>(NU, ring, a.ns, tc, wave, lane, wk, fail, wpre);
        wait_count(&tc->staged, TILE_NWP + TILE_NSTASH, fail);
        const float Sf = bc[0];
        const double sc = scale_of(bc[4]), cA = tile_cA(tc, 0, (double)D);
        tl_stamp(a.tl, 5);
        auto on_tile = [&](int t) {
            if (lane < TH) {
                const int li = t * TH + lane, mi = ch0 + li;
                const long long M = tile_row_sum<TH>(tsum, t, lane);
                const double xold = sd[li];
                const float accf = (float)xold + ((float)(sc * ((double)M + cA + TILE_CU * (double)srs[li])) + Sf);   // f32 accumulator pre-loaded with x (:548)
                const double xnew = (double)accf;                                                                    // :553
                a.x[mi] = xnew;
                a.sxy[so + mi] = sd[CPW + li] * ((xold - mean1) * rstd1) + sd[2 * CPW + li];                         // mixatt's state write (:385): ln1 output
                SitePre<2> pre;
#pragma unroll
                for (int k = 0; k < 3; k++) pre.p[k] = reinterpret_cast<const f32x4 *>(sp + li * 12)[k];
                SiteAcc<2> acc;
                acc.clear();
                site_emit<2>(pre, a.dy, D, mi, xnew, sd[3 * CPW + li], acc);
                tile_site_leave<2>(acc, scr, CPW, li);
            }
        };
        tile_consume<TH, S, UPT, RUN, HEAD>(NU, ring, a.ns, tc, tsum, xq, xvd_t, wave, lane, fail, wk, wpre, [](int) { return 0; }, on_tile, a.tl);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();
    tile_site_publish(a.dy, scr, CPW);
    ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

// ffn_v dequant-GEMV, x += v * sigmoid(r) (rwkv.cu:574-577); commits state dd; opens the next site: TPC tiles of K = 4 D
template <int SD, int S, int KBT, int TH, int TPC, int NVN>
__global__ __launch_bounds__(NT) void k_ffnv_t(FfnVTArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FfnVArgs &a = ta.a;
    constexpr int CPW = TH * TPC, NTILE = TPC, UPT = KBT / S, NU = NTILE * UPT, RUN = tile_run(TH, UPT, TPC), PW = site_pw<NVN>();
    constexpr int HEAD = TH == 16 && RUN == 1 ? RWKV_TILE_HEAD_FV : 0, NE = HEAD / TILE_NSTASH > 0 ? HEAD / TILE_NSTASH : 1;
    static_assert(HEAD <= NU && HEAD % TILE_NSTASH == 0, "the head: whole rounds of the three non-staging waves, inside the stream");
    double *red = reinterpret_cast<double *>(smem);
    const int D = a.D, lane = threadIdx.x & 63, wave = wave_id();
    const int xvd_t = (D >> 2) * 12;                       // the 4 D hidden units as ONE vector
    unsigned *xq = reinterpret_cast<unsigned *>(smem + RED_BYTES);
    // LDS: [scratch][1 staged vector of 4 D][stash: xold, lw, lb, prevn [4][CPW] f64 | P [CPW][16] f32 | rs[CPW] | rgate[CPW]][publish scratch][tile sums][TileCtl][ring]
    double *sd = reinterpret_cast<double *>(xq + xvd_t);
    float *sp = reinterpret_cast<float *>(sd + 4 * CPW);
    unsigned *srs = reinterpret_cast<unsigned *>(sp + CPW * 16);
    float *srg = reinterpret_cast<float *>(srs + CPW);
    double *scr = reinterpret_cast<double *>(srg + CPW);
    int *tsum = reinterpret_cast<int *>(scr + CPW * 10);
    TileCtl *tc = reinterpret_cast<TileCtl *>(tsum + NTILE * TH * 3);
    unsigned char *ring = reinterpret_cast<unsigned char *>(tc + 1);
    RWKV_ARGS_NOW(a.hbuf, a.partS, a.partM, a.n_part);
    const int cb0 = blockIdx.x * TPC, ch0 = blockIdx.x * CPW;
    tl_stamp(a.tl, 0);
    auto unit_src = [&](int u) {
        const int t = u / UPT, c = u - t * UPT;
        return ta.im.bimg + ((size_t)(cb0 + t) * KBT + (size_t)c * S) * 1024;
    };
    unsigned fail = 0u;
    if (wave == NC) {
        fail = tile_loader<S, UPT, RUN, TH, (TH == 16 ? RWKV_TILE_PRE_FV : RWKV_TILE_PRE_FV4), HEAD>(NU, NTILE, ring, a.ns, tc, tsum, lane, unit_src, a.tl);
        tl_stamp(a.tl, 2);
    } else {
        const double mean2 = a.lnstat[0], rstd2 = a.lnstat[1];
        const size_t so = (size_t)a.ctl->slot * a.slot_stride;
        if (wave < TILE_NWP) {
            tile_vec<4 * SD>(a.hbuf, a.partS, a.partM, a.n_part, 4 * D, red, xq, tc, a.tl);
        } else {
            auto dsrc = [&](int i) { const int g = i / CPW, j = i - g * CPW; return g == 0 ? a.x[ch0 + j] : g == 1 ? a.lnw[ch0 + j] : g == 2 ? a.lnb[ch0 + j] : (NVN == 3 ? a.sprev[so + ch0 + j] : 0.0); };
            double d0 = 0.0, d1 = 0.0;
            f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned v0 = 0u;
            constexpr int NP4 = CPW * PW / 4;      // f32x4 pieces of the site's producer constants of the CPW channels
            if (wave == 4) { if (lane < 4 * CPW) d0 = dsrc(lane); if (lane + 64 < 4 * CPW) d1 = dsrc(lane + 64); }
            else if (wave == 5) { if (lane < NP4) p0 = reinterpret_cast<const f32x4 *>(a.st.P + (size_t)ch0 * PW)[lane]; if (lane + 64 < NP4) p1 = reinterpret_cast<const f32x4 *>(a.st.P + (size_t)ch0 * PW)[lane + 64]; }
            else if (lane < 2 * CPW) v0 = lane < CPW ? a.rs[ch0 + lane] : __float_as_uint(a.rgate[ch0 + lane - CPW]);
            TL_BAR(a.tl, 1);
            __syncthreads();   // order
            TL_BAR(a.tl, 2);
            if (wave == 4) { if (lane < 4 * CPW) sd[lane] = d0; if (lane + 64 < 4 * CPW) sd[lane + 64] = d1; }
            else if (wave == 5) { if (lane < NP4) reinterpret_cast<f32x4 *>(sp)[lane] = p0; if (lane + 64 < NP4) reinterpret_cast<f32x4 *>(sp)[lane + 64] = p1; }
            else if (lane < 2 * CPW) srs[lane] = v0;          // (rs[CPW] and rgate[CPW] are adjacent)
            tile_stash_done(tc, lane);
        }
        float *bc = reinterpret_cast<float *>(red + RED_BC);
        TileWalk wk;
        u32x4 wpre[NE][S];
        if (wave >= TILE_NWP) tile_pretake<S, HEAD>(NU, ring, a.ns, tc, wave, lane, wk, fail, wpre);
        wait_count(&tc->staged, TILE_NWP + TILE_NSTASH, fail);
        const float Sf = bc[0];
        const double sc = scale_of(bc[4]), cA = tile_cA(tc, 0, 4.0 * (double)D);
        tl_stamp(a.tl, 5);
        auto on_tile = [&](int t) {
            if (lane < TH) {
                const int li = t * TH + lane, g = ch0 + li;
                const long long M = tile_row_sum<TH>(tsum, t, lane);
                const float v = (float)(sc * ((double)M + cA + TILE_CU * (double)srs[li])) + Sf;
                const double xold = sd[li];
                const double xnew = xold + (double)(v * srg[li]);                         // blockout, rwkv.cu:407 (f32 product)
                a.x[g] = xnew;
                a.sdd[so + g] = sd[CPW + li] * ((xold - mean2) * rstd2) + sd[2 * CPW + li];   // mixffn's state write (:344): ln2 output
                SitePre<NVN> pre;
#pragma unroll
                for (int k = 0; k < PW / 4; k++) pre.p[k] = reinterpret_cast<const f32x4 *>(sp + li * PW)[k];
                SiteAcc<NVN> acc;
                acc.clear();
                site_emit<NVN>(pre, a.dy, D, g, xnew, sd[3 * CPW + li], acc);
                tile_site_leave<NVN>(acc, scr, CPW, li);
            }
        };
        tile_consume<TH, S, UPT, RUN, HEAD>(NU, ring, a.ns, tc, tsum, xq, xvd_t, wave, lane, fail, wk, wpre, [](int) { return 0; }, on_tile, a.tl);
    }
    tl_stamp(a.tl, 6);
    __syncthreads();
    tile_site_publish(a.dy, scr, CPW);
    ring_report(fail, a.herr);
    tl_stamp(a.tl, 7);
}

} // namespace rwkvk
