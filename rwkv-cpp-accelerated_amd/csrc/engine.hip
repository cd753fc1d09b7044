// engine.hip -- host runtime + C-ABI of librwkv_mi355x.so (see include/rwkv_mi355x.h).
//
// Replaces the reference's backend translation unit include/rwkv/cuda/rwkv.cu (load / setState /
// getOutput / freeTensors / cuda_rwkv_parralel, declared rwkv.h:63-122) with an opaque-handle
// engine: weights are re-tiled once at load, state and the embedding table stay resident on the
// device, and one token is a replay of a captured hipGraph (4 launches per layer + 2).
#include "kernels.hip.h"
#include "tile.hip.h"
#include "seq.hip.h"
#include "sampler.hip.h"
#include "../../include/rwkv_mi355x.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

using namespace rwkvk;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(RWKV_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// tensor slots of model.bin, reference enums/enum.h:7-55
enum {
    X, EMBED, LAYERNORMS, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD, BUFFER1, BUFFER2, BUFFER3, BUFFER4,
    MIXK, MIXV, MIXR, KM, VM, RM, KR, VR, RR, O1, O2, O3, ATTOUT, ATTOUTR, ATTOUTO, FFNMIXK, FFNMIXV,
    FFNK, FFNV, FFNR, FFNKR, FFNVR, FFNRR, FFNKO, FFNVO, FFNRO, FFNKBUFFER, FFNVBUFFER, FFNRBUFFER,
    DECAY, BONUS, HEAD, HEADR, HEADO
};
// element sizes / counts: reference rwkv.h:84 and rwkv.h:124-128
const uint64_t kTypes[46] = {8, 4, 8, 8, 8, 8, 8, 8, 8, 4, 4, 4, 8, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4,
                             4, 1, 4, 4, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4, 4, 8, 8, 4, 8, 8, 1, 4, 4};
uint64_t tensor_elems(int i, uint64_t a, uint64_t b)
{
    const uint64_t V = RWKV_VOCAB;
    const uint64_t s[46] = {b, V * b, 4 * (a + 1) * b, a * b, a * b, a * b, a * b, a * b, b, V, b, b,
                            a * b, a * b, a * b, a * b * b, a * b * b, a * b * b, a * b, a * b, a * b, a * b, a * b, a * b,
                            a * b * b, a * b, a * b, a * b, a * b, a * b * b * 4, a * b * b * 4, a * b * b,
                            a * b, a * b * 4, a * b, a * b, a * b * 4, a * b, b, b, b * 4, a * b, a * b, V * b, b, b};
    return s[i];
}

// where tensor bytes come from during a load, and how they reach the device.  A model.bin (8 GB at 7B) streams through
// two pinned staging buffers: the file read of piece i + 1 overlaps the host-to-device copy of piece i, nothing waits for
// the device until the load is complete (round 1 read every tensor into a pageable buffer and synchronised per tensor).
struct Source {
    FILE *f = nullptr;                  // model.bin, or
    const void *const *ptrs = nullptr;  // 46 pointers in file layout
    bool on_device = false;
    uint64_t off[46];
    static constexpr size_t PIN = 32u << 20;
    unsigned char *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int turn = 0;
    // device pointer of a tensor that already lives on the device (on_device sources), else nullptr
    const void *device_ptr(int i, uint64_t o) const { return (ptrs && on_device && ptrs[i]) ? static_cast<const unsigned char *>(ptrs[i]) + o : nullptr; }
    // enqueue bytes [o, o + n) of tensor i -> dst (device) on `stream`; the caller's later work on the stream sees them
    int to_device(int i, uint64_t o, uint64_t n, void *dst, hipStream_t stream)
    {
        if (ptrs) {
            if (!ptrs[i]) return fail(RWKV_E_IO, "tensor slot %d: missing", i);
            // caller-owned memory stays valid for the whole load: no staging, no wait
            HIPCHK(hipMemcpyAsync(dst, static_cast<const unsigned char *>(ptrs[i]) + o, n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
            return 0;
        }
        if (fseeko(f, (off_t)(off[i] + o), SEEK_SET) != 0) return fail(RWKV_E_IO, "tensor slot %d: seek failed", i);
        for (uint64_t done = 0; done < n;) {
            const size_t m = (size_t)std::min<uint64_t>(PIN, n - done);
            const int b = turn;
            turn ^= 1;
            if (!pin[b]) {
                HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&pin[b]), PIN, hipHostMallocDefault));
                HIPCHK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
            }
            if (used[b]) HIPCHK(hipEventSynchronize(ev[b]));          // the copy that last read this buffer has finished
            if (fread(pin[b], 1, m, f) != m) return fail(RWKV_E_IO, "tensor slot %d: short read", i);
            HIPCHK(hipMemcpyAsync(static_cast<unsigned char *>(dst) + done, pin[b], m, hipMemcpyHostToDevice, stream));
            HIPCHK(hipEventRecord(ev[b], stream));
            used[b] = true;
            done += m;
        }
        return 0;
    }
    ~Source()
    {
        for (int b = 0; b < 2; b++) {
            if (used[b]) (void)hipEventSynchronize(ev[b]);
            if (ev[b]) (void)hipEventDestroy(ev[b]);
            if (pin[b]) (void)hipHostFree(pin[b]);
        }
    }
};

} // namespace

// RCCL (librccl.so, resolved at run time by rwkv_pipe_init: a single-GPU user never loads it) -- the subset of rccl.h the
// layer pipeline needs.  Types per /opt/rocm/include/rccl/rccl.h (ncclUniqueId = 128 opaque bytes; ncclUint64 = 5, ncclFloat64 = 8).
struct Pipe {
    void *lib = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1;
    struct Id { char b[128]; };
    int (*GetUniqueId)(Id *) = nullptr;
    int (*CommInitRank)(void **, int, Id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    // two-communicator schedule (rwkv_pipe_decode_dual): sub-schedule k = streams of parity k on comm k (comm / comm2) and HIP stream cs[k],
    // hop buffers per parity, events between the comm streams and the engine's compute stream
    void *comm2 = nullptr;
    hipStream_t cs[2] = {nullptr, nullptr};
    double *xin[2] = {nullptr, nullptr}, *xout[2] = {nullptr, nullptr};
    unsigned long long *idbuf[2] = {nullptr, nullptr};
    hipEvent_t ev_rx[2] = {nullptr, nullptr}, ev_cp[2] = {nullptr, nullptr};
    bool dual_ready = false;     // every resource of the two-communicator schedule AND comm2 exist (pipe_dual_setup)
    uint64_t ch = 0;             // rows per prefill micro-batch every rank of this pipeline agreed on (rwkv_pipe_init)
    std::string info;            // one JSON object describing this rank's end of the transport (rwkv_pipe_info)
};

// decode kernels that stream through the LDS ring: -1 = by model width (measured on MI355X, profiles/r02/ring_sweep.txt,
// profiles/r04/early_take_ab.txt): rows of 3-5 KiB gain 2-5 % with k_att, k_ffn_rk and k_ffnv on the ring; <= 2 KiB rows lose
#ifndef RWKV_RING
#define RWKV_RING -1
#endif

// per-chunk scratch of the chunk path (one set per pipeline stage that may be in flight)
struct SeqScratch {
    double *state = nullptr;            // [D] LayerNorm output of the chunk's last token
    float *y = nullptr;                 // gated wkv output [SEQ_T][D]
    unsigned *img[3] = {nullptr, nullptr, nullptr}, *imgh = nullptr;
    SeqPart *qpart = nullptr, *qparta = nullptr, *qparth = nullptr;
    SeqStat *stat = nullptr;
    float *pk3 = nullptr, *pk5 = nullptr, *pk1 = nullptr;
};

struct rwkv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int grid = 256;          // workgroups per launch = compute units
    bool loaded = false;
    uint64_t L = 0, D = 0, maxT = 1;
    uint64_t l0 = 0, l1 = UINT64_MAX;   // pipeline stage: this context owns layers [l0, l1) (whole model by default)
    int S = 0;               // ceil(D / 1024): 1 KiB row pieces per lane
    int seq_rows = SEQ_TM;   // chunk path: rows per weight pass, 64 (two halves, round 4) or 32 (env RWKV_SEQ_ROWS)
    int seq_b = -1;          // 64-row passes: GEMM kinds (bit 0 K/V/R, bit 2 ffn k/r) that run as k_seq_gemm_b -- one vector's image of the whole slice resident,
                             // a wave's tiles in batches, one round of workgroups (env RWKV_SEQ_B; 0: k_seq_gemm_p<.., true, 2>; -1: ffn k/r always, K/V/R at D >= 4096:
                             // +1-2 % there, -1.5 % at D = 2048, profiles/r04/gemm_b_ab2.txt)
    int tile = -1;           // decode kernel classes that run in TILE form (tile.hip.h; bit 0 k_att, 1 k_attout, 2 k_ffn_rk, 3 k_ffnv; env RWKV_TILE; -1 = auto: 15 at
                             // D = 4096 and 5120 on 256 CUs, 13 at D = 2048, else 0).  15: the context holds ONLY the tile image of the per-layer
                             // matrices; a partial mask keeps both layouts (tuning)
    int ring = RWKV_RING;    // decode kernels that stream their weights through the LDS ring (bit 0 k_att, 1 k_attout, 2 k_ffn_rk, 3 k_ffnv, 4 k_head; env RWKV_RING)

    // weights (device)
    float *embed = nullptr;
    double *ln = nullptr, *mixk = nullptr, *mixv = nullptr, *mixr = nullptr, *fmixk = nullptr, *fmixr = nullptr;
    float *kr = nullptr, *vr = nullptr, *rr = nullptr, *o1 = nullptr, *o2 = nullptr, *o3 = nullptr;
    float *attr = nullptr, *atto = nullptr, *fkr = nullptr, *fvr = nullptr, *frr = nullptr;
    float *fko = nullptr, *fvo = nullptr, *fro = nullptr, *headr = nullptr, *heado = nullptr;
    double *uw = nullptr, *ew = nullptr;
    // LayerNorm-site tables (kernels.hip.h "LayerNorm sites"): k = 0 ln1 -> K/V/R (3 vectors), 1 ln2 -> ffn k/r (2), 2 ln_out -> head (1)
    float *siteC[3] = {nullptr, nullptr, nullptr};    // [L or 1][NV][D]
    float *siteP[3] = {nullptr, nullptr, nullptr};    // [L or 1][D][PW]
    double *siteTC[3] = {nullptr, nullptr, nullptr};  // [L or 1][NV]
    float *siteMC[3] = {nullptr, nullptr, nullptr};   // [L or 1][NV]
    float *siteB[3] = {nullptr, nullptr, nullptr};    // [NV][D]   per-token, emitted by the row owners
    double *sitePD[3] = {nullptr, nullptr, nullptr};  // [grid][8] per-workgroup partial tuples
    float *sitePF[3] = {nullptr, nullptr, nullptr};   // [grid][4]
    double *lnstat = nullptr;                         // [3][2] mean, rstd per site
    uint8_t *w_kvr = nullptr, *w_att = nullptr, *w_frk = nullptr, *w_fv = nullptr, *w_head = nullptr;
    unsigned *rs_kvr = nullptr, *rs_att = nullptr, *rs_frk = nullptr, *rs_fv = nullptr, *rs_head = nullptr;   // row sums
    // state + scratch (device)
    double *state[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double *x = nullptr, *partA = nullptr, *partF = nullptr;
    float *ybuf = nullptr, *hbuf = nullptr, *rgate = nullptr, *logits = nullptr, *blk_val = nullptr, *partMA = nullptr, *partMF = nullptr;
    unsigned *blk_idx = nullptr;
    Ctl *ctl = nullptr;
    Ctl *h_ctl = nullptr;    // pinned staging, maxT entries
    unsigned long long *gen = nullptr;
    unsigned long long *pick = nullptr;      // device: id drawn by the sampler
    double *ts_part = nullptr;               // sampler scratch (sampler.hip.h)
    float *ts_p = nullptr, *ts_pw = nullptr;
    unsigned *ts_key = nullptr;
    unsigned gen_cap = 0;
    hipGraphExec_t g_fwd = nullptr, g_greedy = nullptr;
    // device error word: a mapped pinned word that a kernel raises when one of its bounded waits gave up (LDS ring hand-offs);
    // checked after every stream synchronisation (device_check)
    unsigned *herr = nullptr, *d_herr = nullptr;
    unsigned long long *tl = nullptr;   // phase-timeline buffer (debug), [grid][NW][8]
    bool tl_on = false;
    int tl_cls = 3;                     // kernel class the timeline instruments (env RWKV_TL_CLASS, 1..4)
    // chunked (prompt prefill) path, seq.hip.h: allocated when max_ctx > 1 on a whole-model context
    bool seq_ok = false;
    unsigned long long *sq_tokens = nullptr;      // device [SQ_RING][SEQ_T]: token ids of the chunks in flight
    unsigned long long *h_sq_tokens = nullptr;    // pinned, same shape
    hipEvent_t sq_ev[8] = {};                     // slot r of the ring is free once sq_ev[r] has completed
    uint64_t sq_n = 0;                            // chunks enqueued so far
    // captured passes of the chunk path (GPT mode): one hipGraph per (layer range, residual buffer, rows, logits row of the last part);
    // a pass is 11 launches per layer -- 2.9 k launches and events for a 512-token 7B prompt, 3.5 us of host time each
    // (RWKV_SEQ_GRAPH=0: direct launches)
    struct SeqGraphKey { uint64_t la, lb, row0; int buf, n; bool operator<(const SeqGraphKey &o) const { return std::tie(la, lb, row0, buf, n) < std::tie(o.la, o.lb, o.row0, o.buf, o.n); } };
    std::map<SeqGraphKey, hipGraphExec_t> sq_graphs;
    bool seq_graph = true;
    int graphs = 3;                               // env RWKV_GRAPH: bit 0 = a decode token is a hipGraph replay, bit 1 = so is a GPT-mode pass of the chunk path (0: direct launches)
    double *sq_x[4] = {nullptr, nullptr, nullptr, nullptr};   // residual stream [SEQ_T][D]; two buffers: a pipeline stage receives chunk c + 1 while chunk c is sent on; [2], [3]: more chunks in flight in rwkv_forward's pipeline
    hipEvent_t xs_ev[2] = {nullptr, nullptr};     // rwkv_xseq_copy: "source chunk done" / "copied"
    // long prompts on one GPU: the chunk path as a two-stage software pipeline (layers [l0, mid) on `stream`, [mid, l1) + head on
    // `stream2`, stage 2 on chunk c while stage 1 is on chunk c + 1); the second stage has its own scratch set
    static constexpr int SPLIT_MAX = 4;
    int n_split = 0;                                          // stages in use (0: not set up)
    hipStream_t sp_stream[SPLIT_MAX] = {};                    // [0] = stream
    struct SeqScratch *sp_scratch[SPLIT_MAX] = {};            // [0] = the context's own set (built on the fly)
    hipEvent_t sp_done[SPLIT_MAX][SPLIT_MAX] = {};            // [stage][buffer]: the stage has finished the chunk in that buffer
    hipEvent_t sp_end = nullptr;
    double *x_in = nullptr;                       // decode: residual vector received from the previous stage (nullptr: c->x)
    struct Pipe *pipe = nullptr;                  // RCCL transport of the layer pipeline (rwkv_pipe_init)
    bool pipe_prof = false;                       // rwkv_pipe_profile: bracket the ticks' RCCL groups with event pairs
    static constexpr size_t HOP_EV = 512;
    hipEvent_t hop_ev[2 * HOP_EV] = {};
    double hop_stats[4] = {0.0, 0.0, 0.0, 0.0};
    Ctl *pipe_ring = nullptr;                     // pinned control blocks of rwkv_pipe_decode's items (grown on demand, kept)
    uint64_t pipe_ring_cap = 0;
    double *sq_state = nullptr;                   // [D] LayerNorm output of the chunk's last token
    float *sq_y = nullptr;                        // gated wkv output [SEQ_T][D]
    unsigned *sq_img[3] = {nullptr, nullptr, nullptr}, *sq_imgh = nullptr;   // MFMA A-operand images (K = D; K = 4D)
    SeqPart *sq_qpart = nullptr, *sq_qparta = nullptr, *sq_qparth = nullptr;   // quantisation records: site vectors [3][SEQ_T][SEQ_O], att_out input [SEQ_T][SEQ_O], ffn_v input [SEQ_T][SEQ_O]
    SeqStat *sq_stat = nullptr;                          // [SEQ_T][SEQ_O] LayerNorm partial statistics
    float *sq_pk3 = nullptr, *sq_pk5 = nullptr, *sq_pk1 = nullptr;   // per-slice partial values [SEQ_O][SEQ_T][3D / 5D / D] of the K/V/R, ffn k/r, att_out | ffn_v GEMMs
    // second resident copy of the matrices (chunked path only): MFMA B-operand images, row sums per octant of K
    uint8_t *b_kvr = nullptr, *b_att = nullptr, *b_frk = nullptr, *b_fv = nullptr, *b_head = nullptr;
    // tile images the tile-form DECODE kernels stream (tile.hip.h): the chunk path's own (16-row tiles) where a workgroup owns one 16-channel
    // block, else a decode-only image of 4-row tiles
    uint8_t *t_kvr = nullptr, *t_att = nullptr, *t_frk = nullptr, *t_fv = nullptr;
    int tile_th = 0, tile_s = 0, tile_tpc = 0;     // rows per tile, KiB per ring unit, tiles per row class and workgroup (0: no tile form at this width)
    unsigned *r8_kvr = nullptr, *r8_att = nullptr, *r8_frk = nullptr, *r8_fv = nullptr, *r8_head = nullptr;
    std::vector<void *> allocs;
    size_t alloc_bytes = 0;                       // device bytes behind `allocs` (rwkv_resident_bytes)
};

namespace {

template <typename T> int dalloc(rwkv_ctx *c, T **p, size_t count)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16)));
    c->allocs.push_back(q);
    c->alloc_bytes += std::max<size_t>(count * sizeof(T), 16);
    *p = static_cast<T *>(q);
    return 0;
}

// one staged vector = 3 limb planes of S x 1 KiB
size_t smem_att(int S) { return RED_BYTES + 3 * (size_t)S * 3072; }
size_t smem_attout(int S) { return RED_BYTES + (size_t)S * 3072; }
size_t smem_frk(int S) { return RED_BYTES + 2 * (size_t)S * 3072; }
size_t smem_fv(int S) { return RED_BYTES + 4 * (size_t)S * 3072; }
size_t smem_head(int S) { return RED_BYTES + (size_t)S * 3072 + NW * 8; }
// ring kernels: units of one row (S KiB) behind the staged vectors, as many as fit the CU's 160 KiB (a group of R rows takes R
// consecutive units, wrapping: every group size shares the same ring)
constexpr size_t LDS_BYTES = 160 * 1024;
int ring_slots(size_t fixed, int, int S)
{
    return (int)((LDS_BYTES - fixed - sizeof(GldsCtl)) / ((size_t)S * 1024));
}
size_t smem_ring(size_t fixed, int R, int S) { return fixed + sizeof(GldsCtl) + (size_t)ring_slots(fixed, R, S) * S * 1024; }
// k_att / k_ffn_rk / k_ffnv in ring form: [scratch][nv staged vectors][control block][ring]
int ring_units(int nv, int S) { return (int)((LDS_BYTES - RED_BYTES - sizeof(GldsCtl) - (size_t)nv * S * 3072) / ((size_t)S * 1024)); }
size_t smem_ring3(int nv, int S) { return RED_BYTES + (size_t)nv * S * 3072 + sizeof(GldsCtl) + (size_t)ring_units(nv, S) * S * 1024; }

// tile-form decode kernels (tile.hip.h; classes 1 att, 2 att_out, 3 ffn_rk, 4 ffn_v): ring of S KiB units behind each kernel's fixed LDS.
// Which widths have a tile form: those whose channels split into whole TH-row tiles per workgroup on this grid --
//   D = 4096 on 256 CUs: 16 channels = ONE 16-row tile per class (the chunk path's own image; the default there)
//   D = 5120 on 256 CUs: 20 channels = five 4-row tiles (a decode-only image of 4-row tiles)
//   D = 2048 on 256 CUs:  8 channels = two 4-row tiles
struct TileCfg { int th, s, tpc; };
TileCfg tile_cfg_for(uint64_t D, int grid)
{
    if (grid != 256) return TileCfg{0, 0, 0};
    if (D == 4096) return TileCfg{16, 4, 1};
    if (D == 5120) return TileCfg{4, 5, 5};
    if (D == 2048) return TileCfg{4, 4, 2};
    return TileCfg{0, 0, 0};
}
size_t tile_fixed(const rwkv_ctx *c, int cls)
{
    const int D = (int)c->D, cpw = c->tile_th * c->tile_tpc;
    return cls == 1 ? tile_fixed_att(D, cpw) : cls == 2 ? tile_fixed_attout(D, cpw) : cls == 3 ? tile_fixed_frk(D, cpw) : tile_fixed_fv(D, cpw);
}
int tile_units(const rwkv_ctx *c, int cls)
{
    const int kbt = (int)((cls == 4 ? 4 * c->D : c->D) * c->tile_th / 1024);                         // fragments of a tile along K
    const int nu = (cls == 1 ? 3 : cls == 3 ? 5 : 1) * c->tile_tpc * kbt / c->tile_s;                 // units a workgroup streams
    const int fit = (int)((LDS_BYTES - tile_fixed(c, cls)) / ((size_t)c->tile_s * 1024)) & ~1;         // (even: the loader moves pairs of units)
    return std::min(fit, nu);
}
size_t tile_smem(const rwkv_ctx *c, int cls) { return tile_fixed(c, cls) + (size_t)tile_units(c, cls) * c->tile_s * 1024; }
// a class runs in tile form when asked to (RWKV_TILE bit cls - 1) and its image is there
bool tile_ok(const rwkv_ctx *c, int cls)
{
    const uint8_t *img = cls == 1 ? c->t_kvr : cls == 2 ? c->t_att : cls == 3 ? c->t_frk : c->t_fv;
    return c->tile > 0 && ((c->tile >> (cls - 1)) & 1) && img != nullptr;
}
// launch of a tile-form kernel template for the context's configuration
#define TILE_DISPATCH(c, CALL16, CALL4_5, CALL4_2)                                  \
    do {                                                                            \
        if ((c)->tile_th == 16) { CALL16; }                                         \
        else if ((c)->tile_tpc == 5) { CALL4_5; }                                   \
        else { CALL4_2; }                                                           \
    } while (0)

// k-blocks a wave of k_seq_gemm_p keeps in flight (K/V/R, ffn k/r at up to 4 KiB rows; a divisor of 8)
#ifndef RWKV_SEQ_DEPTH0
#define RWKV_SEQ_DEPTH0 2
#endif
#ifndef RWKV_SEQ_DEPTH2
#define RWKV_SEQ_DEPTH2 2
#endif
#ifndef RWKV_ATTOUT_R
#define RWKV_ATTOUT_R 2
#endif
constexpr int ATTOUT_R = RWKV_ATTOUT_R;
constexpr int SITE_NV[3] = {3, 2, 1};
constexpr int SQ_RING = 8;   // prompt chunks whose token ids may be in flight between host and device
constexpr int SITE_PW[3] = {site_pw<3>(), site_pw<2>(), site_pw<1>()};

#define DISPATCH_S(S, ...)                                           \
    switch (S) {                                                     \
    case 1: { constexpr int S_ = 1; __VA_ARGS__; } break;            \
    case 2: { constexpr int S_ = 2; __VA_ARGS__; } break;            \
    case 3: { constexpr int S_ = 3; __VA_ARGS__; } break;            \
    case 4: { constexpr int S_ = 4; __VA_ARGS__; } break;            \
    default: { constexpr int S_ = 5; __VA_ARGS__; } break;           \
    }

template <typename K> int allow_smem(K kernel, size_t bytes)
{
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}


// ---- argument blocks of the decode kernels ----
// (the stage's first ln1 site is opened by k_first's few workgroups, every other site by the full grid)
struct ArgMaker {
    rwkv_ctx *c;
    int D, grid, n_first;
    size_t LD;
    explicit ArgMaker(rwkv_ctx *c_) : c(c_), D((int)c_->D), grid(c_->grid), LD((size_t)c_->L * c_->D) { n_first = grid < 32 ? grid : 32; }
    unsigned long long *tl_of(int k, uint64_t l) const { return (c->tl_on && k == c->tl_cls && l == (c->l0 + c->l1) / 2) ? c->tl : nullptr; }
    SiteStatic site_static(int k, uint64_t ll) const
    {
        const int nv = SITE_NV[k], pw = SITE_PW[k];
        SiteStatic st;
        st.C = c->siteC[k] + (size_t)ll * nv * D; st.P = c->siteP[k] + (size_t)ll * D * pw;
        st.TC = c->siteTC[k] + (size_t)ll * nv; st.maxC = c->siteMC[k] + (size_t)ll * nv;
        st.invD = 1.0 / (double)D; st.invDm1 = 1.0 / (double)(D - 1);
        return st;
    }
    SiteDyn site_dyn(int k, int n_part) const
    {
        SiteDyn dy;
        dy.B = c->siteB[k]; dy.pd = c->sitePD[k]; dy.pf = c->sitePF[k]; dy.lnstat = c->lnstat + 2 * k; dy.n_part = n_part;
        return dy;
    }
    FirstArgs first() const
    {
        FirstArgs fa;
        fa.embed = c->embed; fa.ln = c->ln; fa.x = c->x; fa.x_in = c->x_in ? c->x_in : c->x; fa.st = site_static(0, c->l0); fa.dy = site_dyn(0, n_first);
        fa.sxy = c->state[0] + (size_t)c->l0 * D; fa.slot_stride = LD; fa.ctl = c->ctl; fa.D = D; fa.from_token = c->l0 == 0;
        return fa;
    }
    AttArgs att(uint64_t l) const
    {
        const size_t lo = (size_t)l * D;
        AttArgs aa;
        aa.x = c->x; aa.st = site_static(0, l); aa.dy = site_dyn(0, l == c->l0 ? n_first : grid);
        aa.w = c->w_kvr ? c->w_kvr + (size_t)(l - c->l0) * 3 * D * D : nullptr; aa.rs = c->rs_kvr + (size_t)(l - c->l0) * D * 3;
        aa.uw = c->uw + lo; aa.ew = c->ew + lo;
        aa.r_att = c->attr + lo; aa.o_att = c->atto + lo;
        aa.saa = c->state[1] + lo; aa.sbb = c->state[2] + lo;
        aa.slot_stride = LD; aa.ybuf = c->ybuf; aa.partS = c->partA; aa.partM = c->partMA;
        aa.ctl = c->ctl; aa.D = D; aa.ns = 0; aa.tl = tl_of(1, l); aa.herr = c->d_herr;
        return aa;
    }
    AttOutArgs attout(uint64_t l) const
    {
        const size_t lo = (size_t)l * D;
        AttOutArgs ao;
        ao.w = c->w_att ? c->w_att + (size_t)(l - c->l0) * D * D : nullptr; ao.rs = c->rs_att + (size_t)(l - c->l0) * D; ao.ybuf = c->ybuf; ao.partS = c->partA; ao.partM = c->partMA; ao.n_part = grid;
        ao.x = c->x; ao.lnw = c->ln + (4 * l + 2) * D; ao.lnb = c->ln + (4 * l + 3) * D; ao.lnstat = c->lnstat + 0;
        ao.sxy = c->state[0] + lo; ao.st = site_static(1, l); ao.dy = site_dyn(1, grid); ao.sdd = c->state[4] + lo;
        ao.slot_stride = LD; ao.ctl = c->ctl; ao.D = D; ao.ns = 0; ao.tl = tl_of(2, l); ao.herr = c->d_herr;
        return ao;
    }
    FfnRKArgs frk(uint64_t l) const
    {
        const size_t lo = (size_t)l * D;
        FfnRKArgs fa;
        fa.x = c->x; fa.st = site_static(1, l); fa.dy = site_dyn(1, grid);
        fa.w = c->w_frk ? c->w_frk + (size_t)(l - c->l0) * 5 * D * D : nullptr; fa.rs = c->rs_frk + (size_t)(l - c->l0) * D * 5;
        fa.r_fv = c->fvr + 4 * lo; fa.o_fv = c->fvo + 4 * lo;
        fa.hbuf = c->hbuf; fa.rgate = c->rgate; fa.partS = c->partF; fa.partM = c->partMF; fa.ctl = c->ctl; fa.D = D;
        fa.ns = 0; fa.tl = tl_of(3, l); fa.herr = c->d_herr;
        return fa;
    }
    // the site ffn_v opens: ln1 of layer l + 1 (3 vectors), or ln_out -> head after the stage's last layer (on a non-final
    // pipeline stage nobody reads it: the next stage re-opens its own site from x)
    bool fv_next_att(uint64_t l) const { return l + 1 < c->l1; }
    FfnVArgs fv(uint64_t l) const
    {
        const size_t lo = (size_t)l * D;
        FfnVArgs fv;
        fv.w = c->w_fv ? c->w_fv + (size_t)(l - c->l0) * 4 * D * D : nullptr; fv.rs = c->rs_fv + (size_t)(l - c->l0) * D; fv.hbuf = c->hbuf; fv.partS = c->partF; fv.partM = c->partMF; fv.n_part = grid;
        fv.rgate = c->rgate; fv.x = c->x; fv.lnw = c->ln + (4 * l + 4) * D; fv.lnb = c->ln + (4 * l + 5) * D; fv.lnstat = c->lnstat + 2;
        fv.sdd = c->state[4] + lo; fv.slot_stride = LD; fv.ctl = c->ctl; fv.D = D; fv.tl = tl_of(4, l);
        fv.ns = 0; fv.herr = c->d_herr;
        if (fv_next_att(l)) { fv.st = site_static(0, l + 1); fv.dy = site_dyn(0, grid); fv.sprev = c->state[0] + lo + D; }
        else { fv.st = site_static(2, 0); fv.dy = site_dyn(2, grid); fv.sprev = nullptr; }
        return fv;
    }
    HeadArgs head() const
    {
        HeadArgs ha;
        ha.x = c->x; ha.st = site_static(2, 0); ha.dy = site_dyn(2, grid); ha.w = c->w_head; ha.rs = c->rs_head; ha.logits = c->logits;
        ha.blk_val = c->blk_val; ha.blk_idx = c->blk_idx; ha.ctl = c->ctl; ha.D = D; ha.ns = 0; ha.herr = c->d_herr;
        return ha;
    }
};

// ---- one launch helper per kernel class (0 embed, 1 att, 2 att_out, 3 ffn_rk, 4 ffn_v, 5 head, 6 argmax) ----
void launch_class(rwkv_ctx *c, int cls, uint64_t l)
{
    const int S = c->S, grid = c->grid;
    const ArgMaker mk(c);
    switch (cls) {
    case 0: {
        FirstArgs fa = mk.first();
        k_first<<<dim3(mk.n_first), dim3(NT), 0, c->stream>>>(fa);
    } break;
    case 1: {
        AttArgs aa = mk.att(l);
        if (tile_ok(c, 1)) {
            AttTArgs ta;
            ta.a = aa; ta.a.ns = tile_units(c, 1);
            ta.im.CB = mk.D / c->tile_th; ta.im.bimg = c->t_kvr + (size_t)(l - c->l0) * 3 * (size_t)mk.D * mk.D;
            const size_t sm = tile_smem(c, 1);
            TILE_DISPATCH(c, (k_att_t<4, 4, 64, 16, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_att_t<5, 5, 20, 4, 5><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_att_t<2, 4, 8, 4, 2><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)));
        }
        else if (c->ring & 1) {
            aa.ns = ring_units(3, S);
            DISPATCH_S(S, k_att<S_, 1, 1><<<dim3(grid), dim3(NT), smem_ring3(3, S), c->stream>>>(aa));
        }
        else DISPATCH_S(S, k_att<S_, nb_att(S_)><<<dim3(grid), dim3(NT), smem_att(S), c->stream>>>(aa));
    } break;
    case 2: {
        AttOutArgs ao = mk.attout(l);
        if (tile_ok(c, 2)) {
            AttOutTArgs ta;
            ta.a = ao; ta.a.ns = tile_units(c, 2);
            ta.im.CB = mk.D / c->tile_th; ta.im.bimg = c->t_att + (size_t)(l - c->l0) * (size_t)mk.D * mk.D;
            const size_t sm = tile_smem(c, 2);
            TILE_DISPATCH(c, (k_attout_t<4, 4, 64, 16, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_attout_t<5, 5, 20, 4, 5><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_attout_t<2, 4, 8, 4, 2><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)));
        }
        else if (c->ring & 2) { ao.ns = ring_slots(smem_attout(S), ATTOUT_R, S); DISPATCH_S(S, k_attout<S_, ATTOUT_R, 1, 1><<<dim3(grid), dim3(NT), smem_ring(smem_attout(S), ATTOUT_R, S), c->stream>>>(ao)); }
        else DISPATCH_S(S, k_attout<S_, ATTOUT_R, nb_attout(S_)><<<dim3(grid), dim3(NT), smem_attout(S), c->stream>>>(ao));
    } break;
    case 3: {
        FfnRKArgs fa = mk.frk(l);
        if (tile_ok(c, 3)) {
            FfnRKTArgs ta;
            ta.a = fa; ta.a.ns = tile_units(c, 3);
            ta.im.CB = mk.D / c->tile_th; ta.im.bimg = c->t_frk + (size_t)(l - c->l0) * 5 * (size_t)mk.D * mk.D;
            const size_t sm = tile_smem(c, 3);
            TILE_DISPATCH(c, (k_ffn_rk_t<4, 4, 64, 16, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_ffn_rk_t<5, 5, 20, 4, 5><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                          (k_ffn_rk_t<2, 4, 8, 4, 2><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)));
        }
        else if (c->ring & 4) {
            fa.ns = ring_units(2, S);
            DISPATCH_S(S, k_ffn_rk<S_, 1, 1><<<dim3(grid), dim3(NT), smem_ring3(2, S), c->stream>>>(fa));
        }
        else DISPATCH_S(S, k_ffn_rk<S_, nb_frk(S_)><<<dim3(grid), dim3(NT), smem_frk(S), c->stream>>>(fa));
    } break;
    case 4: {
        FfnVArgs fv = mk.fv(l);
        fv.ns = ring_units(4, S);
        if (tile_ok(c, 4)) {
            FfnVTArgs ta;
            ta.a = fv; ta.a.ns = tile_units(c, 4);
            ta.im.CB = mk.D / c->tile_th; ta.im.bimg = c->t_fv + (size_t)(l - c->l0) * 4 * (size_t)mk.D * mk.D;
            const size_t sm = tile_smem(c, 4);
            if (mk.fv_next_att(l))
                TILE_DISPATCH(c, (k_ffnv_t<4, 4, 256, 16, 1, 3><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                              (k_ffnv_t<5, 5, 80, 4, 5, 3><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                              (k_ffnv_t<2, 4, 32, 4, 2, 3><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)));
            else
                TILE_DISPATCH(c, (k_ffnv_t<4, 4, 256, 16, 1, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                              (k_ffnv_t<5, 5, 80, 4, 5, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)),
                              (k_ffnv_t<2, 4, 32, 4, 2, 1><<<dim3(grid), dim3(NT), sm, c->stream>>>(ta)));
        }
        else if (mk.fv_next_att(l)) {
            if (c->ring & 8) DISPATCH_S(S, k_ffnv<S_, 3, 1, 1><<<dim3(grid), dim3(NT), smem_ring3(4, S), c->stream>>>(fv))
            else DISPATCH_S(S, k_ffnv<S_, 3, nb_fv(S_)><<<dim3(grid), dim3(NT), smem_fv(S), c->stream>>>(fv));
        } else {
            if (c->ring & 8) DISPATCH_S(S, k_ffnv<S_, 1, 1, 1><<<dim3(grid), dim3(NT), smem_ring3(4, S), c->stream>>>(fv))
            else DISPATCH_S(S, k_ffnv<S_, 1, nb_fv(S_)><<<dim3(grid), dim3(NT), smem_fv(S), c->stream>>>(fv));
        }
    } break;
    case 5: {
        HeadArgs ha = mk.head();
        if (c->ring & 16) { ha.ns = ring_slots(smem_head(S), RWKV_HEAD_RR, S); DISPATCH_S(S, k_head<S_, 1, 1><<<dim3(grid), dim3(NT), smem_ring(smem_head(S), RWKV_HEAD_RR, S), c->stream>>>(ha)); }
        else DISPATCH_S(S, k_head<S_, nb_head(S_)><<<dim3(grid), dim3(NT), smem_head(S), c->stream>>>(ha));
    } break;
    default:
        k_argmax_finish<<<dim3(1), dim3(64), 0, c->stream>>>(c->blk_val, c->blk_idx, grid, c->ctl, c->gen, c->gen_cap);
    }
}

// after a synchronisation: did a bounded wait inside a kernel give up?  (a ring hand-off that never arrived: the GPU is shared
// or preempted, or a workgroup died.)  The token's results are not to be trusted: fail loudly instead of returning them.
int device_check(rwkv_ctx *c)
{
    if (!c->herr || *c->herr == 0u) return 0;
    const unsigned code = *c->herr;
    *c->herr = 0u;
    return fail(RWKV_E_DEVICE, "a device-side wait gave up (code %u: LDS ring hand-off timed out; is the GPU shared or preempted?) -- "
                               "the results of this call are invalid; RWKV_RING=0 selects the register kernels", code);
}

// enqueue the kernels of one token on the context's stream.  ev: optional array of
// (4L + 4) events recorded before each launch and after the last (profiling).
int enqueue_token(rwkv_ctx *c, bool with_argmax, hipEvent_t *ev)
{
    const bool last = c->l1 == c->L;
    int evi = 0;
#define EV() do { if (ev) HIPCHK(hipEventRecord(ev[evi++], c->stream)); } while (0)
    EV();
    launch_class(c, 0, 0);   // first stage: embed + ln0; every stage: open the ln1 site of its first layer
    for (uint64_t l = c->l0; l < c->l1; l++)
        for (int cls = 1; cls <= 4; cls++) { EV(); launch_class(c, cls, l); }
    EV();
    if (last) launch_class(c, 5, 0);
    EV();
    if (last && with_argmax) launch_class(c, 6, 0);
    EV();
#undef EV
    HIPCHK(hipGetLastError());
    return 0;
}

int build_graph(rwkv_ctx *c, bool with_argmax, hipGraphExec_t *out)
{
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    int rc = enqueue_token(c, with_argmax, nullptr);
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc) return rc;
    HIPCHK(e);
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g));
    return 0;
}

// (re)capture the token graphs around the context's current argument blocks (x_in changes with rwkv_pipe_init / rwkv_pipe_free)
int rebuild_graphs(rwkv_ctx *c)
{
    int rc = 0;
    if (c->g_fwd) { (void)hipGraphExecDestroy(c->g_fwd); c->g_fwd = nullptr; rc = build_graph(c, false, &c->g_fwd); if (rc) { c->g_fwd = nullptr; return rc; } }
    if (c->g_greedy) { (void)hipGraphExecDestroy(c->g_greedy); c->g_greedy = nullptr; rc = build_graph(c, true, &c->g_greedy); if (rc) { c->g_greedy = nullptr; return rc; } }
    return 0;
}

int retile(rwkv_ctx *c, Source &src, int slot, uint64_t layer, uint64_t N, uint64_t M, uint8_t *dst,
           int G, int RS, int off, uint8_t *staging)
{
    const uint8_t *dsrc = static_cast<const uint8_t *>(src.device_ptr(slot, layer * N * M));
    if (!dsrc) {
        // `staging` is reused by the next matrix: stream order (this copy is enqueued behind the previous re-tile kernel) protects it
        int rc = src.to_device(slot, layer * N * M, N * M, staging, c->stream);
        if (rc) return rc;
        dsrc = staging;
    }
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
    hipLaunchKernelGGL(k_retile, grid, dim3(256), 0, c->stream, dsrc, dst, (int)N, (int)M, G, RS, off);
    HIPCHK(hipGetLastError());
    return 0;
}

template <typename T> int upload(rwkv_ctx *c, Source &src, int slot, T **dst)
{
    const uint64_t n = tensor_elems(slot, c->L, c->D);
    int rc = dalloc(c, dst, n);
    if (rc) return rc;
    // stream-ordered with the pack / re-tile kernels that read the copy (a default-stream D2D copy
    // would race with them: the engine stream is non-blocking)
    return src.to_device(slot, 0, n * sizeof(T), *dst, c->stream);
}

int seq_smem_limits()
{
    int rc = 0;
#define SEQ_ALLOW_P(TAG, NTW, NKB, NVS, DEPTH, MULTI) if (!rc) rc = allow_smem(k_seq_gemm_p<TAG, NTW, NKB, NVS, DEPTH, MULTI>, seq_gemm_p_smem(NKB, NVS, MULTI))
    SEQ_ALLOW_P(0, 3, 8, 3, RWKV_SEQ_DEPTH0, false); SEQ_ALLOW_P(0, 3, 10, 2, 2, false);
    SEQ_ALLOW_P(1, 1, 8, 1, 8, false); SEQ_ALLOW_P(1, 1, 10, 1, 10, false);
    SEQ_ALLOW_P(2, 5, 8, 2, RWKV_SEQ_DEPTH2, false); SEQ_ALLOW_P(2, 4, 10, 2, 2, false);
    SEQ_ALLOW_P(3, 1, 4, 1, 4, true);
#undef SEQ_ALLOW_P
#define SEQ_ALLOW_P2(TAG, NTW, NKB, NVS, DEPTH, MULTI) if (!rc) rc = allow_smem(k_seq_gemm_p<TAG, NTW, NKB, NVS, DEPTH, MULTI, 2>, seq_gemm_p_smem(NKB, NVS, MULTI, 2))
    SEQ_ALLOW_P2(0, 3, 2, 3, 2, true); SEQ_ALLOW_P2(1, 1, 8, 1, 8, false); SEQ_ALLOW_P2(1, 1, 10, 1, 10, false);
    SEQ_ALLOW_P2(2, 3, 2, 2, 2, true); SEQ_ALLOW_P2(3, 1, 4, 1, 4, true);
#undef SEQ_ALLOW_P2
    if (!rc) rc = allow_smem(k_seq_gemm_b<0, 3, RWKV_SEQ_BDEPTH, 2>, seq_gemm_b_smem(SEQ_B_NKB_MAX, 2));
    if (!rc) rc = allow_smem(k_seq_gemm_b<2, 3, RWKV_SEQ_BDEPTH, 2>, seq_gemm_b_smem(SEQ_B_NKB_MAX, 2));
    if (!rc) rc = allow_smem(k_seq_gemm_ks, SEQ_KS_SMEM);
    return rc;
}

int set_smem_limits(rwkv_ctx *c)
{
    const int S = c->S;
    int rc = 0;
    DISPATCH_S(S, rc = allow_smem(k_att<S_, nb_att(S_)>, smem_att(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_attout<S_, ATTOUT_R, nb_attout(S_)>, smem_attout(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffn_rk<S_, nb_frk(S_)>, smem_frk(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffnv<S_, 3, nb_fv(S_)>, smem_fv(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffnv<S_, 1, nb_fv(S_)>, smem_fv(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_head<S_, nb_head(S_)>, smem_head(S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_att<S_, 1, 1>, smem_ring3(3, S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_attout<S_, ATTOUT_R, 1, 1>, smem_ring(smem_attout(S), ATTOUT_R, S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffn_rk<S_, 1, 1>, smem_ring3(2, S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffnv<S_, 3, 1, 1>, smem_ring3(4, S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_ffnv<S_, 1, 1, 1>, smem_ring3(4, S))); if (rc) return rc;
    DISPATCH_S(S, rc = allow_smem(k_head<S_, 1, 1>, smem_ring(smem_head(S), RWKV_HEAD_RR, S))); if (rc) return rc;
    if (c->tile_th) {
#define TILE_ALLOW(K16, K45, K42, CLS)                                                                   \
        do {                                                                                             \
            if (c->tile_th == 16) rc = allow_smem(K16, tile_smem(c, CLS));                               \
            else if (c->tile_tpc == 5) rc = allow_smem(K45, tile_smem(c, CLS));                          \
            else rc = allow_smem(K42, tile_smem(c, CLS));                                                \
            if (rc) return rc;                                                                           \
        } while (0)
        TILE_ALLOW((k_att_t<4, 4, 64, 16, 1>), (k_att_t<5, 5, 20, 4, 5>), (k_att_t<2, 4, 8, 4, 2>), 1);
        TILE_ALLOW((k_attout_t<4, 4, 64, 16, 1>), (k_attout_t<5, 5, 20, 4, 5>), (k_attout_t<2, 4, 8, 4, 2>), 2);
        TILE_ALLOW((k_ffn_rk_t<4, 4, 64, 16, 1>), (k_ffn_rk_t<5, 5, 20, 4, 5>), (k_ffn_rk_t<2, 4, 8, 4, 2>), 3);
        TILE_ALLOW((k_ffnv_t<4, 4, 256, 16, 1, 3>), (k_ffnv_t<5, 5, 80, 4, 5, 3>), (k_ffnv_t<2, 4, 32, 4, 2, 3>), 4);
        TILE_ALLOW((k_ffnv_t<4, 4, 256, 16, 1, 1>), (k_ffnv_t<5, 5, 80, 4, 5, 1>), (k_ffnv_t<2, 4, 32, 4, 2, 1>), 4);
#undef TILE_ALLOW
    }
    return 0;
}

int load_common(rwkv_ctx *c, Source &src, uint64_t L, uint64_t D, uint64_t max_ctx)
{
    if (c->loaded) return fail(RWKV_E_STATE, "RWKV already loaded");   // reference: rwkv.h:283-286
    if (L == 0 || D == 0 || D % 16 != 0 || D > 5120 || L > 4096)
        return fail(RWKV_E_ARG, "unsupported model shape n_layers=%llu n_embed=%llu (n_embed must be a multiple of 16, <= 5120)",
                    (unsigned long long)L, (unsigned long long)D);
    if (max_ctx == 0) max_ctx = 1;
    if ((uint64_t)c->grid * 512 < D)      // every decode kernel hands a workgroup its share of the D channels in groups of <= 512
        return fail(RWKV_E_ARG, "RWKV_GRID=%d is too small for n_embed=%llu (need >= %llu workgroups)", c->grid, (unsigned long long)D, (unsigned long long)((D + 511) / 512));
    HIPCHK(hipSetDevice(c->device));
    c->L = L; c->D = D; c->maxT = max_ctx; c->S = (int)((D + 1023) / 1024);
    if (c->ring < 0) c->ring = c->S >= 3 ? 13 : 0;      // (5 KiB rows: k_ffnv joined the ring in round 4, with the early take: 21.4 -> 20.8 us, profiles/r04/early_take_ab.txt)
    if (c->l1 == UINT64_MAX) c->l1 = L;
    if (c->l0 >= c->l1 || c->l1 > L) return fail(RWKV_E_ARG, "layer range [%llu, %llu) does not fit a %llu-layer model", (unsigned long long)c->l0, (unsigned long long)c->l1, (unsigned long long)L);
    const uint64_t l0 = c->l0, l1 = c->l1, nl = l1 - l0;
    const bool first = l0 == 0, last = l1 == L;
    const uint64_t V = RWKV_VOCAB;
    int rc;

    // vectors: as-is
    if (first && (rc = upload(c, src, EMBED, &c->embed))) return rc;   // the table lives on the first stage only
    if ((rc = upload(c, src, LAYERNORMS, &c->ln))) return rc;
    if ((rc = upload(c, src, MIXK, &c->mixk))) return rc;
    if ((rc = upload(c, src, MIXV, &c->mixv))) return rc;
    if ((rc = upload(c, src, MIXR, &c->mixr))) return rc;
    if ((rc = upload(c, src, KR, &c->kr))) return rc;
    if ((rc = upload(c, src, VR, &c->vr))) return rc;
    if ((rc = upload(c, src, RR, &c->rr))) return rc;
    if ((rc = upload(c, src, O1, &c->o1))) return rc;
    if ((rc = upload(c, src, O2, &c->o2))) return rc;
    if ((rc = upload(c, src, O3, &c->o3))) return rc;
    if ((rc = upload(c, src, ATTOUTR, &c->attr))) return rc;
    if ((rc = upload(c, src, ATTOUTO, &c->atto))) return rc;
    if ((rc = upload(c, src, FFNMIXK, &c->fmixk))) return rc;
    if ((rc = upload(c, src, FFNMIXV, &c->fmixr))) return rc;   // slot named "v" holds time_mix_r
    if ((rc = upload(c, src, FFNKR, &c->fkr))) return rc;
    if ((rc = upload(c, src, FFNVR, &c->fvr))) return rc;
    if ((rc = upload(c, src, FFNRR, &c->frr))) return rc;
    if ((rc = upload(c, src, FFNKO, &c->fko))) return rc;
    if ((rc = upload(c, src, FFNVO, &c->fvo))) return rc;
    if ((rc = upload(c, src, FFNRO, &c->fro))) return rc;
    if ((rc = upload(c, src, HEADR, &c->headr))) return rc;
    if ((rc = upload(c, src, HEADO, &c->heado))) return rc;
    double *decay = nullptr, *bonus = nullptr;
    if ((rc = upload(c, src, DECAY, &decay))) return rc;
    if ((rc = upload(c, src, BONUS, &bonus))) return rc;
    if ((rc = dalloc(c, &c->uw, L * D))) return rc;
    if ((rc = dalloc(c, &c->ew, L * D))) return rc;
    hipLaunchKernelGGL(k_prep_wkv, dim3((unsigned)((L * D + 255) / 256)), dim3(256), 0, c->stream, decay, bonus, c->uw, c->ew, (size_t)(L * D));

    // LayerNorm-site tables for the layers of this stage (+ the head site)
    for (int k = 0; k < 3; k++) {
        const uint64_t n = k == 2 ? 1 : L;
        if ((rc = dalloc(c, &c->siteC[k], n * SITE_NV[k] * D))) return rc;
        if ((rc = dalloc(c, &c->siteP[k], n * D * SITE_PW[k]))) return rc;
        if ((rc = dalloc(c, &c->siteTC[k], n * SITE_NV[k]))) return rc;
        if ((rc = dalloc(c, &c->siteMC[k], n * SITE_NV[k]))) return rc;
        HIPCHK(hipMemsetAsync(c->siteP[k], 0, n * D * SITE_PW[k] * sizeof(float), c->stream));
    }
    {
        auto build = [&](int k, uint64_t ll, int m, const double *lnw, const double *lnb, const double *mix, const float *r, const float *o) {
            k_site_static<<<dim3(1), dim3(NT), 0, c->stream>>>(lnw, lnb, mix, r, o, c->siteC[k] + (size_t)ll * SITE_NV[k] * D,
                                                               c->siteP[k] + (size_t)ll * D * SITE_PW[k], c->siteTC[k] + (size_t)ll * SITE_NV[k],
                                                               c->siteMC[k] + (size_t)ll * SITE_NV[k], m, SITE_PW[k], (int)D);
        };
        for (uint64_t l = l0; l < l1; l++) {
            const size_t lo = (size_t)l * D;
            const double *w1 = c->ln + (4 * l + 2) * D, *b1 = c->ln + (4 * l + 3) * D, *w2 = c->ln + (4 * l + 4) * D, *b2 = c->ln + (4 * l + 5) * D;
            build(0, l, 0, w1, b1, c->mixk + lo, c->kr + lo, c->o1 + lo);
            build(0, l, 1, w1, b1, c->mixv + lo, c->vr + lo, c->o2 + lo);
            build(0, l, 2, w1, b1, c->mixr + lo, c->rr + lo, c->o3 + lo);
            build(1, l, 0, w2, b2, c->fmixk + lo, c->fkr + lo, c->fko + lo);
            build(1, l, 1, w2, b2, c->fmixr + lo, c->frr + lo, c->fro + lo);
        }
        build(2, 0, 0, c->ln + (4 * L + 2) * D, c->ln + (4 * L + 3) * D, nullptr, c->headr, c->heado);
        HIPCHK(hipGetLastError());
    }

    // uint8 matrices.  Two device layouts exist: ROW form -- re-tiled to row-per-output, what the row-form decode kernels
    // stream -- and the TILE image -- [16-row tile][k-block of 64][lane][16 B], signed: the MFMA B operand of the chunk path AND what the
    // tile-form decode kernels stream (tile.hip.h).  A decode class (K/V/R, att_out, ffn k/r, ffn_v: bits 0..3 of c->tile) is resident in
    // the ONE layout its decode kernel streams: the matrices of a class that runs in tile form go file layout -> row form in a scratch
    // buffer -> row sums, octant row sums, tile image, layer by layer, and their row form is never kept (7B-wide models on 256 CUs: all
    // four classes).  The 16-row tile image of every class is also what the chunk path (max_ctx > 1) multiplies with; at the widths whose
    // decode tiles are 4 rows high (tile.hip.h) a tile-form class has its own image.  The head stays in row form (k_head) either way
    // (+ its tile image for the chunk path).
    const bool want_seq = [&] { const char *e = getenv("RWKV_SEQ"); return max_ctx > 1 && D % 64 == 0 && !(e && e[0] == '0'); }();
    const TileCfg tcfg = tile_cfg_for(D, c->grid);
    // auto: every class where a workgroup owns exactly one 16-channel block (D = 4096) and, on 4-row tiles, at D = 5120; K/V/R, ffn k/r and ffn_v at
    // D = 2048 (att_out: 4.8 us in row form, 5.2 in tile form).  Measured per mask with round 6's per-class heads and matrix-core consumers
    // (profiles/r06/tile_masks_ab.txt: 14B 345.7 -> 365 tokens/s, 1B5 1345 -> 1417); round 5's kernels had every class but 14B's ffn k/r faster in row form.
    if (c->tile < 0) c->tile = tcfg.th == 16 || tcfg.tpc == 5 ? 15 : tcfg.tpc == 2 ? 13 : 0;
    if (tcfg.th == 0) c->tile = 0;
    c->tile &= 15;
    if (c->tile) { c->tile_th = tcfg.th; c->tile_s = tcfg.s; c->tile_tpc = tcfg.tpc; }
    auto in_tile = [&](int cls) { return ((c->tile >> (cls - 1)) & 1) != 0; };
    const bool own_t = tcfg.th != 16;                         // the decode kernels' tile image is not the chunk path's
    auto need_b = [&](int cls) { return want_seq || (in_tile(cls) && !own_t); };
    if (!in_tile(1) && (rc = dalloc(c, &c->w_kvr, nl * 3 * D * D))) return rc;
    if (!in_tile(2) && (rc = dalloc(c, &c->w_att, nl * D * D))) return rc;
    if (!in_tile(3) && (rc = dalloc(c, &c->w_frk, nl * 5 * D * D))) return rc;
    if (!in_tile(4) && (rc = dalloc(c, &c->w_fv, nl * 4 * D * D))) return rc;
    if (last && (rc = dalloc(c, &c->w_head, V * D))) return rc;
    if (!rc) rc = dalloc(c, &c->rs_kvr, nl * 3 * D);
    if (!rc) rc = dalloc(c, &c->rs_att, nl * D);
    if (!rc) rc = dalloc(c, &c->rs_frk, nl * 5 * D);
    if (!rc) rc = dalloc(c, &c->rs_fv, nl * D);
    if (!rc) rc = dalloc(c, &c->rs_head, V);
    if (rc) return rc;
    // tile images (+ octant row sums for the chunk path): per = bytes of one layer's image
    const uint64_t cbD = (D + 15) / 16;
    const uint64_t per_kvr = 3 * cbD * 16 * D, per_att = cbD * 16 * D, per_frk = 5 * cbD * 16 * D, per_fv = cbD * 16 * 4 * D;
    if (need_b(1) && (rc = dalloc(c, &c->b_kvr, nl * per_kvr))) return rc;
    if (need_b(2) && (rc = dalloc(c, &c->b_att, nl * per_att))) return rc;
    if (need_b(3) && (rc = dalloc(c, &c->b_frk, nl * per_frk))) return rc;
    if (need_b(4) && (rc = dalloc(c, &c->b_fv, nl * per_fv))) return rc;
    if (want_seq) {
        if ((rc = dalloc(c, &c->r8_kvr, nl * SEQ_O * 3 * D))) return rc;
        if ((rc = dalloc(c, &c->r8_att, nl * SEQ_O * D))) return rc;
        if ((rc = dalloc(c, &c->r8_frk, nl * SEQ_O * 5 * D))) return rc;
        if ((rc = dalloc(c, &c->r8_fv, nl * SEQ_O * D))) return rc;
    }
    if (own_t) {
        if (in_tile(1) && (rc = dalloc(c, &c->t_kvr, nl * 3 * D * D))) return rc;
        if (in_tile(2) && (rc = dalloc(c, &c->t_att, nl * D * D))) return rc;
        if (in_tile(3) && (rc = dalloc(c, &c->t_frk, nl * 5 * D * D))) return rc;
        if (in_tile(4) && (rc = dalloc(c, &c->t_fv, nl * 4 * D * D))) return rc;
    } else {
        if (in_tile(1)) c->t_kvr = c->b_kvr;
        if (in_tile(2)) c->t_att = c->b_att;
        if (in_tile(3)) c->t_frk = c->b_frk;
        if (in_tile(4)) c->t_fv = c->b_fv;
    }
    auto rowsum = [&](const uint8_t *w, unsigned *rs, uint64_t rows, uint64_t N) {
        k_rowsum<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream>>>(w, rs, (size_t)rows, (int)N);
    };
    // one layer's matrix in row form (w_t[N][K]) -> its tile image (and octant row sums)
    auto image = [&](const uint8_t *w_t, uint8_t *bdst, unsigned *r8dst, uint64_t N, uint64_t K, int Q, uint64_t per, int TH = 16) {
        k_bimage<<<dim3((unsigned)((per / 16 + 255) / 256)), dim3(256), 0, c->stream>>>(w_t, bdst, (int)N, (int)K, Q, (int)((((N + Q - 1) / Q) + TH - 1) / TH), TH);
        if (r8dst) k_rowsum8<<<dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream>>>(w_t, r8dst, (int)N, (int)K);
    };
    // load-time scratch (not part of the context): freed on every way out of this function, behind the stream's work
    struct Scratch {
        hipStream_t st; uint8_t *p = nullptr;
        ~Scratch() { if (p) { (void)hipStreamSynchronize(st); (void)hipFree(p); } }
    } staging_s{c->stream}, rowtmp_s{c->stream};
    if (!src.on_device) HIPCHK(hipMalloc(reinterpret_cast<void **>(&staging_s.p), std::max<uint64_t>(4 * D * D, V * D)));
    if (c->tile) HIPCHK(hipMalloc(reinterpret_cast<void **>(&rowtmp_s.p), 5 * D * D));
    uint8_t *const staging = staging_s.p, *const rowtmp = rowtmp_s.p;
    for (uint64_t l = l0; l < l1 && !rc; l++) {
        const uint64_t lr = l - l0;
        uint8_t *kvr = in_tile(1) ? rowtmp : c->w_kvr + lr * 3 * D * D;
        if (!rc) rc = retile(c, src, KM, l, D, D, kvr, 1, 3, 0, staging);
        if (!rc) rc = retile(c, src, VM, l, D, D, kvr, 1, 3, 1, staging);
        if (!rc) rc = retile(c, src, RM, l, D, D, kvr, 1, 3, 2, staging);
        if (!rc) {
            rowsum(kvr, c->rs_kvr + lr * 3 * D, 3 * D, D);
            if (need_b(1)) image(kvr, c->b_kvr + lr * per_kvr, want_seq ? c->r8_kvr + lr * SEQ_O * 3 * D : nullptr, 3 * D, D, 3, per_kvr);
            if (own_t && in_tile(1)) image(kvr, c->t_kvr + lr * 3 * D * D, nullptr, 3 * D, D, 3, 3 * D * D, tcfg.th);
        }
        uint8_t *att = in_tile(2) ? rowtmp : c->w_att + lr * D * D;
        if (!rc) rc = retile(c, src, ATTOUT, l, D, D, att, 1, 1, 0, staging);
        if (!rc) {
            rowsum(att, c->rs_att + lr * D, D, D);
            if (need_b(2)) image(att, c->b_att + lr * per_att, want_seq ? c->r8_att + lr * SEQ_O * D : nullptr, D, D, 1, per_att);
            if (own_t && in_tile(2)) image(att, c->t_att + lr * D * D, nullptr, D, D, 1, D * D, tcfg.th);
        }
        uint8_t *frk = in_tile(3) ? rowtmp : c->w_frk + lr * 5 * D * D;
        if (!rc) rc = retile(c, src, FFNK, l, D, 4 * D, frk, 4, 5, 0, staging);
        if (!rc) rc = retile(c, src, FFNR, l, D, D, frk, 1, 5, 4, staging);
        if (!rc) {
            rowsum(frk, c->rs_frk + lr * 5 * D, 5 * D, D);
            if (need_b(3)) image(frk, c->b_frk + lr * per_frk, want_seq ? c->r8_frk + lr * SEQ_O * 5 * D : nullptr, 5 * D, D, 5, per_frk);
            if (own_t && in_tile(3)) image(frk, c->t_frk + lr * 5 * D * D, nullptr, 5 * D, D, 5, 5 * D * D, tcfg.th);
        }
        uint8_t *fvm = in_tile(4) ? rowtmp : c->w_fv + lr * 4 * D * D;
        if (!rc) rc = retile(c, src, FFNV, l, 4 * D, D, fvm, 1, 1, 0, staging);
        if (!rc) {
            rowsum(fvm, c->rs_fv + lr * D, D, 4 * D);
            if (need_b(4)) image(fvm, c->b_fv + lr * per_fv, want_seq ? c->r8_fv + lr * SEQ_O * D : nullptr, D, 4 * D, 1, per_fv);
            if (own_t && in_tile(4)) image(fvm, c->t_fv + lr * 4 * D * D, nullptr, D, 4 * D, 1, 4 * D * D, tcfg.th);
        }
    }
    if (!rc && last) {
        rc = retile(c, src, HEAD, 0, D, V, c->w_head, 1, 1, 0, staging);
        if (!rc) rowsum(c->w_head, c->rs_head, V, D);
        if (!rc && want_seq) {
            const uint64_t cbV = (V + 15) / 16, per_head = cbV * 16 * D;
            rc = dalloc(c, &c->b_head, per_head);
            if (!rc) rc = dalloc(c, &c->r8_head, SEQ_O * V);
            if (!rc) image(c->w_head, c->b_head, c->r8_head, V, D, 1, per_head);
        }
    }
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());

    // state (zero, as `new RWKVState` does: rwkv.h:163-170) and scratch
    for (int s = 0; s < 5; s++) {
        if ((rc = dalloc(c, &c->state[s], max_ctx * L * D))) return rc;
        HIPCHK(hipMemsetAsync(c->state[s], 0, max_ctx * L * D * sizeof(double), c->stream));
    }
    if ((rc = dalloc(c, &c->x, D))) return rc;
    for (int k = 0; k < 3; k++) {
        if ((rc = dalloc(c, &c->siteB[k], (size_t)SITE_NV[k] * D))) return rc;
        if ((rc = dalloc(c, &c->sitePD[k], (size_t)c->grid * 8))) return rc;
        if ((rc = dalloc(c, &c->sitePF[k], (size_t)c->grid * 4))) return rc;
        HIPCHK(hipMemsetAsync(c->sitePD[k], 0, (size_t)c->grid * 8 * sizeof(double), c->stream));
        HIPCHK(hipMemsetAsync(c->sitePF[k], 0, (size_t)c->grid * 4 * sizeof(float), c->stream));
    }
    if ((rc = dalloc(c, &c->lnstat, 6))) return rc;
    if ((rc = dalloc(c, &c->ybuf, D))) return rc;
    if ((rc = dalloc(c, &c->hbuf, 4 * D))) return rc;
    if ((rc = dalloc(c, &c->rgate, D))) return rc;
    if ((rc = dalloc(c, &c->partA, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->partF, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->partMA, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->partMF, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->blk_val, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->blk_idx, (size_t)c->grid))) return rc;
    if ((rc = dalloc(c, &c->logits, max_ctx * V))) return rc;
    HIPCHK(hipMemsetAsync(c->logits, 0, max_ctx * V * sizeof(float), c->stream));
    if ((rc = dalloc(c, &c->ctl, 1))) return rc;
    HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), c->stream));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_ctl), sizeof(Ctl) * max_ctx, hipHostMallocDefault));
    c->gen_cap = 1u << 16;
    if ((rc = dalloc(c, &c->gen, (size_t)c->gen_cap))) return rc;
    HIPCHK(hipMemsetAsync(c->gen, 0, (size_t)c->gen_cap * sizeof(unsigned long long), c->stream));
    if ((rc = dalloc(c, &c->pick, 1))) return rc;
    if ((rc = dalloc(c, &c->ts_part, (size_t)TS_G * 3))) return rc;
    if ((rc = dalloc(c, &c->ts_p, (size_t)TS_NT * TS_PER))) return rc;
    if ((rc = dalloc(c, &c->ts_pw, (size_t)TS_NT * TS_PER))) return rc;
    if ((rc = dalloc(c, &c->ts_key, (size_t)TS_NT * TS_PER))) return rc;

    // chunked path scratch
    {
        if (want_seq) {
            if ((rc = dalloc(c, &c->sq_tokens, (size_t)SQ_RING * SEQ_TM))) return rc;
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->h_sq_tokens), sizeof(unsigned long long) * SQ_RING * SEQ_TM, hipHostMallocDefault));
            for (int r = 0; r < SQ_RING; r++) HIPCHK(hipEventCreateWithFlags(&c->sq_ev[r], hipEventDisableTiming));
            // (everything per chunk is sized for a pass of SEQ_TM = 64 rows = two halves: half 1's images / records / partial values sit
            // right behind half 0's, seq.hip.h SEQ_TM)
            if ((rc = dalloc(c, &c->sq_x[0], (size_t)SEQ_TM * D))) return rc;
            if ((rc = dalloc(c, &c->sq_x[1], (size_t)SEQ_TM * D))) return rc;
            if ((rc = dalloc(c, &c->sq_state, (size_t)D))) return rc;
            if ((rc = dalloc(c, &c->sq_y, (size_t)SEQ_TM * D))) return rc;
            for (int k = 0; k < 3; k++) {
                if ((rc = dalloc(c, &c->sq_img[k], 2 * a_image_bytes(D) / 4))) return rc;
                HIPCHK(hipMemsetAsync(c->sq_img[k], 0, 2 * a_image_bytes(D), c->stream));
            }
            if ((rc = dalloc(c, &c->sq_imgh, 2 * a_image_bytes(4 * D) / 4))) return rc;
            HIPCHK(hipMemsetAsync(c->sq_imgh, 0, 2 * a_image_bytes(4 * D), c->stream));
            if ((rc = dalloc(c, &c->sq_qpart, (size_t)2 * 3 * SEQ_T * SEQ_O))) return rc;
            if ((rc = dalloc(c, &c->sq_qparta, (size_t)2 * SEQ_T * SEQ_O))) return rc;
            if ((rc = dalloc(c, &c->sq_qparth, (size_t)2 * SEQ_T * SEQ_O))) return rc;
            HIPCHK(hipMemsetAsync(c->sq_qpart, 0, sizeof(SeqPart) * 2 * 3 * SEQ_T * SEQ_O, c->stream));
            HIPCHK(hipMemsetAsync(c->sq_qparta, 0, sizeof(SeqPart) * 2 * SEQ_T * SEQ_O, c->stream));
            HIPCHK(hipMemsetAsync(c->sq_qparth, 0, sizeof(SeqPart) * 2 * SEQ_T * SEQ_O, c->stream));
            if ((rc = dalloc(c, &c->sq_stat, (size_t)SEQ_TM * SEQ_O))) return rc;
            {   // accumulator images: [half][slice][tile][2][4][64] floats, tiles = classes x 16-channel blocks
                const size_t cbd = ((size_t)D + 15) / 16;
                if ((rc = dalloc(c, &c->sq_pk3, (size_t)2 * SEQ_O * 3 * cbd * 512))) return rc;
                if ((rc = dalloc(c, &c->sq_pk5, (size_t)2 * SEQ_O * 5 * cbd * 512))) return rc;
                if ((rc = dalloc(c, &c->sq_pk1, (size_t)2 * SEQ_O * cbd * 512))) return rc;
            }
            c->seq_ok = true;
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = set_smem_limits(c))) return rc;
    if (c->seq_ok && (rc = seq_smem_limits())) return rc;
    if (c->graphs & 1) {
        if ((rc = build_graph(c, false, &c->g_fwd))) return rc;
        if ((rc = build_graph(c, true, &c->g_greedy))) return rc;
    }
    c->loaded = true;
    return 0;
}

// One pass (n <= SEQ_TM = 64 rows: one or two halves of <= 32) through the MFMA path (seq.hip.h) for THIS context's layers [l0, l1); logits rows
// [row0, row0 + n) on the stage that holds the head.
// par == false: GPT-mode semantics of rwkv.cu:493-593 -- n tokens of one sequence, state slot 0, token
// shift along the chunk.  par == true: PARRALEL mode (rwkv.cu:236-240) -- n independent sequences, one
// token each, row t uses state slot row0 + t: the batched decode step, weights read once for all.
// The residual stream of the chunk lives in sq_x[buf]: the first stage fills it from the embedding table, a later
// pipeline stage finds the previous stage's output there (placed by an RCCL recv or a copy) and every stage leaves its
// own output in it.  Nothing here waits for the device: token ids go through a ring of pinned slots.
// part: nullptr = the context's whole layer range on its stream with its own scratch; else layers [la, lb) on part->st with
// scratch part->S (the two-stage software pipeline of rwkv_forward: the embedding belongs to the part that starts at l0, the
// head to the part that ends at l1)
struct ChunkPart { uint64_t la, lb; hipStream_t st; const SeqScratch *S; bool no_upload = false; };
// a pass's token ids: host -> pinned ring slot -> the device slot of the pass's residual buffer (what k_seq_embed reads; the copy is
// stream-ordered behind the embedding kernel of the pass that used the buffer before)
int upload_chunk_tokens(rwkv_ctx *c, const uint64_t *tokens, int n, int buf, hipStream_t st)
{
    const int slot = (int)(c->sq_n % SQ_RING);
    if (c->sq_n >= SQ_RING) HIPCHK(hipEventSynchronize(c->sq_ev[slot]));   // the copy that last used this pinned slot is done
    unsigned long long *h = c->h_sq_tokens + (size_t)slot * SEQ_TM, *d = c->sq_tokens + (size_t)buf * SEQ_TM;
    for (int t = 0; t < n; t++) h[t] = tokens[t];
    HIPCHK(hipMemcpyAsync(d, h, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(c->sq_ev[slot], st));
    c->sq_n++;
    return 0;
}
int enqueue_chunk(rwkv_ctx *c, const uint64_t *tokens, int n, uint64_t row0, bool par, int buf = 0, const ChunkPart *part = nullptr)
{
    const int D = (int)c->D;
    const uint64_t L = c->L, V = RWKV_VOCAB;
    const uint64_t la = part ? part->la : c->l0, lb = part ? part->lb : c->l1;
    const bool first = c->l0 == 0 && la == c->l0, last = c->l1 == c->L && lb == c->l1;
    hipStream_t st = part ? part->st : c->stream;
    SeqScratch own;
    own.state = c->sq_state; own.y = c->sq_y; own.imgh = c->sq_imgh;
    for (int k = 0; k < 3; k++) own.img[k] = c->sq_img[k];
    own.qpart = c->sq_qpart; own.qparta = c->sq_qparta; own.qparth = c->sq_qparth; own.stat = c->sq_stat;
    own.pk3 = c->sq_pk3; own.pk5 = c->sq_pk5; own.pk1 = c->sq_pk1;
    const SeqScratch &S = part ? *part->S : own;
    double *x = c->sq_x[buf];
    if (first) {
        if (buf < 0 || buf >= SQ_RING) return fail(RWKV_E_ARG, "residual buffer %d out of range", buf);
        if (!(part && part->no_upload)) { const int rcu = upload_chunk_tokens(c, tokens, n, buf, st); if (rcu) return rcu; }
        unsigned long long *d = c->sq_tokens + (size_t)buf * SEQ_TM;
        SeqEmbedArgs ea{c->embed, c->ln, d, x, D};
        k_seq_embed<<<dim3(n), dim3(NT), 0, st>>>(ea);
    }
    const size_t LD = (size_t)L * D;
    const int egrid = n * SEQ_O;
    // a pass of 33 .. 64 rows has two halves (seq.hip.h SEQ_TM): the GEMMs read the weights ONCE for both, everything per half sits
    // at these element offsets behind half 0's
    const bool two = n > SEQ_T;
    const size_t cbd_ = ((size_t)D + 15) / 16;
    const size_t h_img = a_image_bytes(D) / 4, h_imgh = a_image_bytes(4 * (size_t)D) / 4;                  // 32-bit words
    const size_t h_part3 = (size_t)3 * SEQ_T * SEQ_O, h_part1 = (size_t)SEQ_T * SEQ_O;                     // records
    const size_t h_pk3 = (size_t)SEQ_O * 3 * cbd_ * 512, h_pk5 = (size_t)SEQ_O * 5 * cbd_ * 512, h_pk1 = (size_t)SEQ_O * cbd_ * 512;   // floats
    const bool big = (D >> 6) > 8 * SEQ_O;       // octants of K = D longer than 8 k-blocks (D > 4096): the NKB = 10 instances
    static const int v012[5] = {0, 1, 2, 0, 0}, v0[5] = {0, 0, 0, 0, 0}, v00001[5] = {0, 0, 0, 0, 1};
    bool tl_layer = false;     // debug timeline (rwkv_debug_timeline with RWKV_TL_CLASS = 10 + GEMM kind): the middle layer's GEMM stamps its phases
    // kind 0 K/V/R, 1 att_out, 2 ffn k/r, 3 ffn_v: tile-per-wave GEMM over the 8 K-slices, partial values into pk
    auto gemm = [&](int kind, const uint8_t *bimg, const unsigned *rs8, int N, int K, int Q, const int *voq, unsigned *const *img, const SeqPart *qpart,
                    float *pk, double *state_dst, size_t gh_img, size_t gh_part, size_t gh_pk) {
        SeqGemmArgs g{};
        g.bimg = reinterpret_cast<const u32x4 *>(bimg); g.rs8 = rs8; g.N = N; g.K = K; g.Q = Q;
        for (int q = 0; q < 5; q++) g.vec_of_q[q] = q < Q ? voq[q] : voq[Q - 1];
        for (int k = 0; k < 3; k++) g.img[k] = reinterpret_cast<const u32x4 *>(img[k]);
        g.part = qpart; g.pk = pk; g.out = nullptr; g.T = n;
        g.img_h = gh_img / 4; g.part_h = gh_part; g.pk_h = gh_pk;      // (images: 16-byte units)
        g.tl = (c->tl_on && c->tl_cls == 10 + kind && tl_layer) ? c->tl : nullptr;
        g.cp_src = S.state; g.cp_dst = state_dst; g.cp_n = (state_dst && !par) ? D : 0;   // GPT: commit the site's state behind it
        const int nch = (N + Q - 1) / Q, ntiles = Q * ((nch + 15) / 16);
        const int ntw_max = kind == 0 ? 3 : kind == 2 ? (two ? 3 : big ? 4 : 5) : 1;      // (two halves: twice the accumulators per weight tile)
        const int RB = (ntiles + SEQ_NW * ntw_max - 1) / (SEQ_NW * ntw_max);
        g.ntw = (ntiles + SEQ_NW * RB - 1) / (SEQ_NW * RB);
        const dim3 grid(SEQ_O * RB), blk(SEQ_NT);
#define SEQ_LAUNCH_P(TAG, NTW, NKB, NVS, DEPTH, MULTI) k_seq_gemm_p<TAG, NTW, NKB, NVS, DEPTH, MULTI><<<grid, blk, seq_gemm_p_smem(NKB, NVS, MULTI), st>>>(g)
#define SEQ_LAUNCH_P2(TAG, NTW, NKB, NVS, DEPTH, MULTI) k_seq_gemm_p<TAG, NTW, NKB, NVS, DEPTH, MULTI, 2><<<grid, blk, seq_gemm_p_smem(NKB, NVS, MULTI, 2), st>>>(g)
        if (two && (kind == 0 || kind == 2) && ((c->seq_b >= 0 ? c->seq_b : (4 | (D >= 4096 ? 1 : 0))) & (1 << kind))) {
            // k_seq_gemm_b: one vector's image of the slice resident, a wave's tiles in batches of <= 3.  The 32 workgroups of a slice are
            // shared out to the matrix's VECTOR GROUPS (runs of row classes on the same vector) so that no workgroup straddles one and the
            // largest tile count of a wave is as small as it gets.  Its conditions: slices of DEPTH .. SEQ_B_NKB_MAX k-blocks (the
            // resident image; row sums prefetched one batch ahead), at most three batches per wave, at most three groups
            const int KB = K >> 6, nkb_min = KB / SEQ_O, nkb_max = (KB + SEQ_O - 1) / SEQ_O, CBt = (nch + 15) / 16;
            SeqGemmBArgs b{};
            int ngrp = 0, gq[4] = {0, 0, 0, 0};
            for (int q = 0; q < Q && ngrp < 4; q++)
                if (q == 0 || voq[q] != voq[q - 1]) gq[ngrp++] = q;
            bool okb = ngrp <= 3 && nkb_min >= RWKV_SEQ_BDEPTH && nkb_max <= SEQ_B_NKB_MAX;
            if (okb) {
                int T[3] = {0, 0, 0}, best[3] = {0, 0, 0};
                for (int i = 0; i < ngrp; i++) { b.grp_tile[i] = gq[i] * CBt; T[i] = ((i + 1 < ngrp ? gq[i + 1] : Q) - gq[i]) * CBt; }
                b.grp_tile[ngrp] = ntiles;
                const int WG = 32;                 // workgroups per slice: 8 slices x 32 = one per CU
                double best_cost = 1e30;
                for (int w0 = 1; w0 <= WG; w0++)
                    for (int w1 = (ngrp > 1 ? 1 : 0); w0 + w1 <= WG; w1 += 1) {
                        const int w2 = ngrp > 2 ? WG - w0 - w1 : 0;
                        if (ngrp == 1 && (w1 || w0 != WG)) continue;
                        if (ngrp == 2 && w0 + w1 != WG) continue;
                        if (ngrp == 3 && w2 < 1) continue;
                        const int w[3] = {w0, w1, w2};
                        double cost = 0;
                        for (int i = 0; i < ngrp; i++) {
                            const int per_wave = (T[i] + SEQ_NW * w[i] - 1) / (SEQ_NW * w[i]);
                            const double c2 = per_wave * 1000.0 + (double)T[i] / (SEQ_NW * w[i]);
                            if (c2 > cost) cost = c2;
                        }
                        if (cost < best_cost) { best_cost = cost; for (int i = 0; i < 3; i++) best[i] = w[i]; }
                        if (ngrp == 1) break;
                    }
                int rb0 = 0, per_wave_max = 0;
                for (int i = 0; i < ngrp; i++) {
                    b.grp_rb[i] = rb0; rb0 += best[i];
                    const int pw = (T[i] + SEQ_NW * best[i] - 1) / (SEQ_NW * best[i]);
                    if (pw > per_wave_max) per_wave_max = pw;
                }
                b.grp_rb[ngrp] = rb0; b.ngrp = ngrp;
                okb = per_wave_max <= 9 && rb0 >= 1;
                if (okb) {
                    b.g = g; b.g.ntw = per_wave_max;
                    const dim3 gridb(SEQ_O * rb0);
                    const size_t smem = seq_gemm_b_smem(nkb_max, 2);
                    if (kind == 0) k_seq_gemm_b<0, 3, RWKV_SEQ_BDEPTH, 2><<<gridb, blk, smem, st>>>(b);
                    else k_seq_gemm_b<2, 3, RWKV_SEQ_BDEPTH, 2><<<gridb, blk, smem, st>>>(b);
                    return;
                }
            }
        }
        if (two) {      // both halves per weight fragment: short k-block groups re-staged into the other LDS buffer (the image of two halves is twice as large)
            if (kind == 0) SEQ_LAUNCH_P2(0, 3, 2, 3, 2, true);
            else if (kind == 1) { if (big) SEQ_LAUNCH_P2(1, 1, 10, 1, 10, false); else SEQ_LAUNCH_P2(1, 1, 8, 1, 8, false); }
            else if (kind == 2) SEQ_LAUNCH_P2(2, 3, 2, 2, 2, true);
            else SEQ_LAUNCH_P2(3, 1, 4, 1, 4, true);
            return;
        }
#undef SEQ_LAUNCH_P2
        // passes of <= 32 rows: ffn_v's activation image (K = 4 D) is staged per 4 k-blocks into alternating LDS buffers instead of per slice, so that
        // the first MFMA does not wait for the whole slice's image (3.36 -> 3.30 ms per 7B chunk; for the other kinds it measured +-0 or a loss:
        // profiles/r04/seq_small_ab.txt)
        if (kind == 3) { SEQ_LAUNCH_P(3, 1, 4, 1, 4, true); return; }
        // the GEMM as a software pipeline over k-blocks, the slice's whole image resident (seq.hip.h k_seq_gemm_p)
        if (kind == 0) { if (big) SEQ_LAUNCH_P(0, 3, 10, 2, 2, false); else SEQ_LAUNCH_P(0, 3, 8, 3, RWKV_SEQ_DEPTH0, false); }
        else if (kind == 1) { if (big) SEQ_LAUNCH_P(1, 1, 10, 1, 10, false); else SEQ_LAUNCH_P(1, 1, 8, 1, 8, false); }
        else { if (big) SEQ_LAUNCH_P(2, 4, 10, 2, 2, false); else SEQ_LAUNCH_P(2, 5, 8, 2, RWKV_SEQ_DEPTH2, false); }
#undef SEQ_LAUNCH_P
    };
    auto resid = [&](int mode, const SeqPart *qpart) {
        SeqResidArgs r{};
        r.x = x; r.pk = S.pk1; r.qpart = qpart; r.pk_gate = S.pk5; r.qpart_gate = S.qpart + (size_t)1 * SEQ_T * SEQ_O;   // ffn r = vector 1 of the ln2 site
        r.stat = S.stat; r.D = D; r.T = n;
        r.pk_h = h_pk1; r.pkg_h = h_pk5; r.part_h = h_part1; r.partg_h = h_part3;
        if (mode == 0) k_seq_resid<0><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(r);
        else if (mode == 1) k_seq_resid<1><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(r);
        else k_seq_resid<2><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(r);
    };
    auto site = [&](int nv, const double *lnw, const double *lnb, const double *const *mix, const float *const *r, const float *const *o, double *state) {
        SeqSiteArgs s{};
        s.x = x; s.stat = S.stat; s.lnw = lnw; s.lnb = lnb;
        for (int q = 0; q < nv; q++) { s.mix[q] = mix ? mix[q] : nullptr; s.r[q] = r[q]; s.o[q] = o[q]; }
        s.state = state; s.state_new = (state && !par) ? S.state : nullptr;
        s.par = par && state; s.state_par = state; s.slot_stride = LD; s.slot0 = (int)row0;
        for (int q = 0; q < 3; q++) s.img[q] = S.img[q];
        s.part = S.qpart; s.D = D; s.T = n;
        s.img_h = h_img; s.part_h = h_part3;
        if (nv == 3) k_seq_site<3><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(s);
        else if (nv == 2) k_seq_site<2><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(s);
        else k_seq_site<1><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(s);
    };
    unsigned *imgh3[3] = {S.imgh, S.imgh, S.imgh};
    const int n_wkv = (D + WKV_CH - 1) / WKV_CH;
    const uint64_t CBd = ((uint64_t)D + 15) / 16;
    resid(0, nullptr);     // LayerNorm statistics of the incoming residual stream (embedding rows, or the previous stage's output)
    for (uint64_t l = la; l < lb; l++) {
        tl_layer = l == (c->l0 + c->l1) / 2;
        const size_t lo = (size_t)l * D, wl = (size_t)(l - c->l0);   // vectors are indexed by the model's layer, matrices by the stage's
        {   // time mix
            const double *mix[3] = {c->mixk + lo, c->mixv + lo, c->mixr + lo};
            const float *r[3] = {c->kr + lo, c->vr + lo, c->rr + lo}, *o[3] = {c->o1 + lo, c->o2 + lo, c->o3 + lo};
            site(3, c->ln + (4 * l + 2) * D, c->ln + (4 * l + 3) * D, mix, r, o, c->state[0] + lo);
            gemm(0, c->b_kvr + wl * 3 * CBd * 16 * D, c->r8_kvr + wl * SEQ_O * 3 * (size_t)D, 3 * D, D, 3, v012, S.img, S.qpart, S.pk3, c->state[0] + lo, h_img, h_part3, h_pk3);
            SeqWkvArgs wa{S.pk3, S.qpart, c->uw + lo, c->ew + lo, c->state[1] + lo, c->state[2] + lo, S.y, D, n, par ? 1 : 0, LD, (int)row0, h_pk3, h_part3};
            if (two) k_seq_wkv<SEQ_TM><<<dim3(n_wkv), dim3(SEQ_TM * WKV_CH), 0, st>>>(wa);
            else k_seq_wkv<SEQ_T><<<dim3(n_wkv), dim3(SEQ_T * WKV_CH), 0, st>>>(wa);
            SeqStageArgs sa{S.y, nullptr, nullptr, c->attr + lo, c->atto + lo, S.img[0], S.qparta, D, n, 0, 0, h_img, h_part1};
            k_seq_stage<0><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(sa);
            gemm(1, c->b_att + wl * CBd * 16 * D, c->r8_att + wl * SEQ_O * (size_t)D, D, D, 1, v0, S.img, S.qparta, S.pk1, nullptr, h_img, h_part1, h_pk1);
            // x = f32(x) + att_out; statistics for ln2.  (Round 3 tried this launch and the site behind it as ONE launch whose (row, octant)
            // workgroups meet on a per-row arrival counter: +3.6 us per fused launch, profiles/r03/prefill_fuse.txt -- the in-launch
            // all-to-all costs more than the kernel boundary it replaces.)
            resid(1, S.qparta);
        }
        {   // channel mix
            const double *mix[3] = {c->fmixk + lo, c->fmixr + lo, nullptr};
            const float *r[3] = {c->fkr + lo, c->frr + lo, nullptr}, *o[3] = {c->fko + lo, c->fro + lo, nullptr};
            site(2, c->ln + (4 * l + 4) * D, c->ln + (4 * l + 5) * D, mix, r, o, c->state[4] + lo);
            gemm(2, c->b_frk + wl * 5 * CBd * 16 * D, c->r8_frk + wl * SEQ_O * 5 * (size_t)D, 5 * D, D, 5, v00001, S.img, S.qpart, S.pk5, c->state[4] + lo, h_img, h_part3, h_pk5);
            SeqStageArgs sh{nullptr, S.pk5, S.qpart, c->fvr + 4 * lo, c->fvo + 4 * lo, S.imgh, S.qparth, 4 * D, n, h_pk5, h_part3, h_imgh, h_part1};
            k_seq_stage<1><<<dim3(egrid), dim3(SEQ_ENT), 0, st>>>(sh);
            gemm(3, c->b_fv + wl * CBd * 16 * 4 * D, c->r8_fv + wl * SEQ_O * (size_t)D, D, 4 * D, 1, v0, imgh3, S.qparth, S.pk1, nullptr, h_imgh, h_part1, h_pk1);
            resid(2, S.qparth);     // x += ffn_v * sigmoid(r); statistics for the next site
        }
    }
    if (last) {   // ln_out and the head
        const float *r[3] = {c->headr, nullptr, nullptr}, *o[3] = {c->heado, nullptr, nullptr};
        site(1, c->ln + (4 * L + 2) * D, c->ln + (4 * L + 3) * D, nullptr, r, o, nullptr);
        SeqGemmArgs g{};
        g.bimg = reinterpret_cast<const u32x4 *>(c->b_head); g.rs8 = c->r8_head; g.N = (int)V; g.K = D; g.Q = 1;
        for (int k = 0; k < 3; k++) g.img[k] = reinterpret_cast<const u32x4 *>(S.img[k]);
        g.part = S.qpart; g.out = c->logits + row0 * V; g.T = two ? SEQ_T : n;
        k_seq_gemm_ks<<<dim3(c->grid), dim3(SEQ_NT), SEQ_KS_SMEM, st>>>(g);
        if (two) {      // the head once per half (its weights are 3 % of a pass's bytes; 240 accumulator registers would not fit one wave)
            for (int k = 0; k < 3; k++) g.img[k] = reinterpret_cast<const u32x4 *>(S.img[k] + h_img);
            g.part = S.qpart + h_part3; g.out = c->logits + (row0 + SEQ_T) * V; g.T = n - SEQ_T;
            k_seq_gemm_ks<<<dim3(c->grid), dim3(SEQ_NT), SEQ_KS_SMEM, st>>>(g);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the chunk path's software pipeline on one GPU: streams, events, one more scratch set and residual-stream buffer per extra
// stage (allocated at first use: 40 MB per stage at 7B).  RWKV_SEQ_STAGES = 1 .. 4 (default 3; 1 = the one-stream schedule)
int split_setup(rwkv_ctx *c)
{
    if (c->n_split) return 0;
    int want = 3;
    if (const char *e = getenv("RWKV_SEQ_STAGES")) want = atoi(e);
    if (want > rwkv_ctx::SPLIT_MAX) want = rwkv_ctx::SPLIT_MAX;
    if ((uint64_t)want > c->L) want = (int)c->L;
    if (want < 2 || !c->seq_ok || c->l0 != 0 || c->l1 != c->L) { c->n_split = 1; return 0; }
    const uint64_t D = c->D;
    c->sp_stream[0] = c->stream;
    HIPCHK(hipEventCreateWithFlags(&c->sp_end, hipEventDisableTiming));
    for (int k = 0; k < want; k++) {
        if (k > 0) HIPCHK(hipStreamCreateWithFlags(&c->sp_stream[k], hipStreamNonBlocking));
        for (int b = 0; b < want; b++) HIPCHK(hipEventCreateWithFlags(&c->sp_done[k][b], hipEventDisableTiming));
    }
    int rc = 0;
    for (int k = 2; k < want && !rc; k++) rc = dalloc(c, &c->sq_x[k], (size_t)SEQ_TM * D);
    for (int k = 1; k < want && !rc; k++) {
        SeqScratch *S = new SeqScratch();
        c->sp_scratch[k] = S;
        if (!rc) rc = dalloc(c, &S->state, (size_t)D);
        if (!rc) rc = dalloc(c, &S->y, (size_t)SEQ_TM * D);
        for (int q = 0; q < 3 && !rc; q++) {
            rc = dalloc(c, &S->img[q], 2 * a_image_bytes(D) / 4);
            if (!rc) HIPCHK(hipMemsetAsync(S->img[q], 0, 2 * a_image_bytes(D), c->stream));
        }
        if (!rc) rc = dalloc(c, &S->imgh, 2 * a_image_bytes(4 * D) / 4);
        if (!rc) HIPCHK(hipMemsetAsync(S->imgh, 0, 2 * a_image_bytes(4 * D), c->stream));
        if (!rc) rc = dalloc(c, &S->qpart, (size_t)2 * 3 * SEQ_T * SEQ_O);
        if (!rc) rc = dalloc(c, &S->qparta, (size_t)2 * SEQ_T * SEQ_O);
        if (!rc) rc = dalloc(c, &S->qparth, (size_t)2 * SEQ_T * SEQ_O);
        if (!rc) {
            HIPCHK(hipMemsetAsync(S->qpart, 0, sizeof(SeqPart) * 2 * 3 * SEQ_T * SEQ_O, c->stream));
            HIPCHK(hipMemsetAsync(S->qparta, 0, sizeof(SeqPart) * 2 * SEQ_T * SEQ_O, c->stream));
            HIPCHK(hipMemsetAsync(S->qparth, 0, sizeof(SeqPart) * 2 * SEQ_T * SEQ_O, c->stream));
        }
        if (!rc) rc = dalloc(c, &S->stat, (size_t)SEQ_TM * SEQ_O);
        const size_t cbd = ((size_t)D + 15) / 16;
        if (!rc) rc = dalloc(c, &S->pk3, (size_t)2 * SEQ_O * 3 * cbd * 512);
        if (!rc) rc = dalloc(c, &S->pk5, (size_t)2 * SEQ_O * 5 * cbd * 512);
        if (!rc) rc = dalloc(c, &S->pk1, (size_t)2 * SEQ_O * cbd * 512);
    }
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_split = want;
    return 0;
}
// first layer of stage k of n: equal shares of the weights, the head's GEMM counting as V / (13 D) layers
uint64_t split_point(const rwkv_ctx *c, int k, int n)
{
    if (k <= 0) return 0;
    if (k >= n) return c->L;
    const double head = (double)RWKV_VOCAB / (13.0 * (double)c->D);
    uint64_t l = (uint64_t)(((double)c->L + head) * k / n + 0.5);
    if (l < (uint64_t)k) l = (uint64_t)k;                                   // every stage holds at least one layer
    if (l > c->L - (uint64_t)(n - k)) l = c->L - (uint64_t)(n - k);
    return l;
}

// The captured passes are keyed by (layer range, residual buffer, rows, logits row of the last part): a server with many different
// prompt lengths keeps adding shapes (up to max_ctx / 64 row offsets x 64 remainders for the part that holds the head).  The cache is
// bounded: beyond SQ_GRAPH_CAP entries everything is dropped and re-captured on demand (a capture costs ~1 ms per pass shape; a 512-token
// prompt uses 24).  Called at the START of rwkv_forward only (the one entry point that replays them): it is synchronous on return, so none of
// the executables is in flight here.
constexpr size_t SQ_GRAPH_CAP = 256;
void trim_pass_graphs(rwkv_ctx *c)
{
    if (c->sq_graphs.size() <= SQ_GRAPH_CAP) return;
    for (auto &kv : c->sq_graphs) (void)hipGraphExecDestroy(kv.second);
    c->sq_graphs.clear();
}

// one pass of the chunk path (GPT mode) over layers [part.la, part.lb) on part.st: replay of the captured graph of exactly this pass
// shape, captured at first use.  The token upload stays outside (host memory changes per pass); everything else a pass launches
// depends only on the key.  A capture that fails turns the graphs off for the context and the pass is launched directly.
int enqueue_pass(rwkv_ctx *c, const uint64_t *tokens, int n, uint64_t row0, int buf, ChunkPart part)
{
    const bool first = c->l0 == 0 && part.la == c->l0, last = c->l1 == c->L && part.lb == c->l1;
    if (!c->seq_graph || c->tl_on) return enqueue_chunk(c, tokens, n, row0, false, buf, &part);
    if (first) { const int rcu = upload_chunk_tokens(c, tokens, n, buf, part.st); if (rcu) return rcu; }
    part.no_upload = true;
    const rwkv_ctx::SeqGraphKey key{part.la, part.lb, last ? row0 : 0, buf, n};
    auto it = c->sq_graphs.find(key);
    if (it == c->sq_graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        bool ok = hipStreamBeginCapture(part.st, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (ok) {
            const int rc = enqueue_chunk(c, nullptr, n, row0, false, buf, &part);
            const hipError_t e = hipStreamEndCapture(part.st, &g);
            ok = rc == 0 && e == hipSuccess && g != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        if (!ok) {
            (void)hipGetLastError();
            c->seq_graph = false;
            return enqueue_chunk(c, nullptr, n, row0, false, buf, &part);
        }
        it = c->sq_graphs.emplace(key, ge).first;
    }
    HIPCHK(hipGraphLaunch(it->second, part.st));
    return 0;
}

// Start of every entry point that launches kernels: the device error word belongs to ONE call.  An entry point that left early
// (a failed HIP call) never consumed it, and a code raised then would surface -- attributed to the wrong operation -- from whatever
// synchronises next; so it is cleared here.
int begin_call(rwkv_ctx *c)
{
    if (c->herr) *c->herr = 0u;
    return 0;
}

int run_token(rwkv_ctx *c, bool with_argmax)
{
    hipGraphExec_t g = with_argmax ? c->g_greedy : c->g_fwd;
    if (g) { HIPCHK(hipGraphLaunch(g, c->stream)); return 0; }
    return enqueue_token(c, with_argmax, nullptr);
}

} // namespace

extern "C" {

const char *rwkv_last_error(void) { return g_err.c_str(); }

int rwkv_create(rwkv_ctx **out, int device)
{
    if (!out) return fail(RWKV_E_ARG, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RWKV_E_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= n) return fail(RWKV_E_ARG, "device %d out of range (have %d)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    rwkv_ctx *c = new rwkv_ctx();
    c->device = device;
    c->grid = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const char *g = getenv("RWKV_GRID");
    if (g && atoi(g) > 0) c->grid = atoi(g);
    { const char *e = getenv("RWKV_RING"); if (e) c->ring = atoi(e); }
    { const char *e = getenv("RWKV_SEQ_B"); if (e) c->seq_b = atoi(e); }
    { const char *e = getenv("RWKV_GRAPH"); if (e) { c->graphs = atoi(e); c->seq_graph = (c->graphs & 2) != 0; } }
    { const char *e = getenv("RWKV_SEQ_ROWS"); if (e) c->seq_rows = atoi(e) > SEQ_T ? SEQ_TM : SEQ_T; }
    { const char *e = getenv("RWKV_TILE"); if (e) c->tile = atoi(e); }
    if (c->grid > NT / 2) c->grid = NT / 2;   // consumers sum one partial per thread of the prologue waves (half the workgroup)
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(RWKV_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    void *dp = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&c->herr), 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, c->herr, 0) != hipSuccess) {
        if (c->herr) (void)hipHostFree(c->herr);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(RWKV_E_DEVICE, "cannot map the device error word");
    }
    *c->herr = 0u;
    c->d_herr = static_cast<unsigned *>(dp);
    *out = c;
    return 0;
}

int rwkv_set_layer_range(rwkv_ctx *c, uint64_t l0, uint64_t l1)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (c->loaded) return fail(RWKV_E_STATE, "RWKV already loaded");
    if (l0 >= l1) return fail(RWKV_E_ARG, "empty layer range");
    c->l0 = l0; c->l1 = l1;
    return 0;
}

int rwkv_load_file(rwkv_ctx *c, const char *path, uint64_t max_ctx)
{
    if (!c || !path) return fail(RWKV_E_ARG, "NULL argument");
    Source src;
    src.f = fopen(path, "rb");
    if (!src.f) return fail(RWKV_E_IO, "Error opening file %s", path);   // reference: rwkv.cu:641-645
    uint64_t hdr[2];
    if (fread(hdr, 8, 2, src.f) != 2) { fclose(src.f); return fail(RWKV_E_IO, "%s: truncated header", path); }
    uint64_t o = 16;
    for (int i = 0; i < 46; i++) { src.off[i] = o; o += tensor_elems(i, hdr[0], hdr[1]) * kTypes[i]; }
    int rc = 0;
    if (fseeko(src.f, 0, SEEK_END) != 0 || (uint64_t)ftello(src.f) < o)
        rc = fail(RWKV_E_IO, "%s: file shorter than the %llu bytes a %llu-layer, %llu-wide model needs", path,
                  (unsigned long long)o, (unsigned long long)hdr[0], (unsigned long long)hdr[1]);
    if (!rc) rc = load_common(c, src, hdr[0], hdr[1], max_ctx);
    fclose(src.f);
    return rc;
}

int rwkv_load_tensors(rwkv_ctx *c, uint64_t n_layers, uint64_t n_embed, const void *const *ptrs,
                      int on_device, uint64_t max_ctx)
{
    if (!c || !ptrs) return fail(RWKV_E_ARG, "NULL argument");
    Source src;
    src.ptrs = ptrs;
    src.on_device = on_device != 0;
    return load_common(c, src, n_layers, n_embed, max_ctx);
}

uint64_t rwkv_n_layers(const rwkv_ctx *c) { return c ? c->L : 0; }
uint64_t rwkv_n_embed(const rwkv_ctx *c) { return c ? c->D : 0; }
uint64_t rwkv_max_ctx(const rwkv_ctx *c) { return c ? c->maxT : 0; }

int rwkv_forward(rwkv_ctx *c, const uint64_t *tokens, uint64_t T, int mode)
{
    if (!c || !tokens) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");                                   // rwkv.h:342-345
    if (T > c->maxT) return fail(RWKV_E_ARG, "Context too large, max context is %llu", (unsigned long long)c->maxT);   // rwkv.h:347-350
    if (mode != RWKV_MODE_PARRALEL && mode != RWKV_MODE_GPT) return fail(RWKV_E_ARG, "bad mode %d", mode);
    for (uint64_t t = 0; t < T; t++)
        if (tokens[t] >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id %llu out of range", (unsigned long long)tokens[t]);
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    trim_pass_graphs(c);
    if (T >= 2 && c->seq_ok && c->l0 == 0 && c->l1 == c->L) {   // prompt chunks (GPT) / batched decode step of T streams (PARRALEL): weights read once per <= 32 rows
        // rows per weight pass: 64 (two halves sharing every weight fragment) for calls of more than 32 rows, else 32
        const uint64_t CH = (T > (uint64_t)SEQ_T && c->seq_rows > SEQ_T) ? (uint64_t)SEQ_TM : (uint64_t)SEQ_T;
        const uint64_t nchunks = (T + CH - 1) / CH;
        int rc = 0;
        if (nchunks >= 2 && (rc = split_setup(c)) == 0 && c->n_split >= 2) {
            // Software pipeline over the chunks: stage k = an equal share of the layers (the last one with the head) on its
            // own stream with its own scratch, stage k on chunk i while stage k - 1 is on chunk i + 1.  Every launch of this path
            // costs ~4.5 us of start-up and tail whatever it moves; with independent kernel sequences on the GPU those run under the
            // other stages' streams (7B, 512-token prompt: 9.1k -> 12k tokens/s with three stages).  A chunk's residual stream stays
            // in its buffer (sq_x[i % n], updated in place by every stage); results are bit-identical to the one-stream schedule.
            const int ns = c->n_split;
            SeqScratch own;      // stage 0 uses the context's own scratch
            own.state = c->sq_state; own.y = c->sq_y; own.imgh = c->sq_imgh;
            for (int k = 0; k < 3; k++) own.img[k] = c->sq_img[k];
            own.qpart = c->sq_qpart; own.qparta = c->sq_qparta; own.qparth = c->sq_qparth; own.stat = c->sq_stat;
            own.pk3 = c->sq_pk3; own.pk5 = c->sq_pk5; own.pk1 = c->sq_pk1;
            HIPCHK(hipEventRecord(c->sp_end, c->stream));                 // the other stages start behind whatever the context's stream holds
            for (int k = 1; k < ns; k++) HIPCHK(hipStreamWaitEvent(c->sp_stream[k], c->sp_end, 0));
            uint64_t i = 0;
            for (uint64_t t0 = 0; t0 < T && !rc; t0 += CH, i++) {
                const int n = (int)(T - t0 < CH ? T - t0 : CH);
                const int b = (int)(i % (uint64_t)ns);
                for (int k = 0; k < ns && !rc; k++) {
                    hipStream_t st = c->sp_stream[k];
                    if (k == 0) { if (i >= (uint64_t)ns) HIPCHK(hipStreamWaitEvent(st, c->sp_done[ns - 1][b], 0)); }   // the last stage is done with this buffer (chunk i - ns)
                    else HIPCHK(hipStreamWaitEvent(st, c->sp_done[k - 1][b], 0));                                        // the stage before has handed chunk i over
                    const ChunkPart part{split_point(c, k, ns), split_point(c, k + 1, ns), st, k == 0 ? &own : c->sp_scratch[k]};
                    if (mode == RWKV_MODE_PARRALEL) rc = enqueue_chunk(c, k == 0 ? tokens + t0 : nullptr, n, t0, true, b, &part);
                    else rc = enqueue_pass(c, k == 0 ? tokens + t0 : nullptr, n, t0, b, part);
                    HIPCHK(hipEventRecord(c->sp_done[k][b], st));
                }
            }
            for (int k = 1; k < ns; k++) {                                // the context's stream owns the result again
                HIPCHK(hipEventRecord(c->sp_end, c->sp_stream[k]));
                HIPCHK(hipStreamWaitEvent(c->stream, c->sp_end, 0));
            }
            if (rc) return rc;
        } else {
            if (rc) return rc;
            SeqScratch own;
            own.state = c->sq_state; own.y = c->sq_y; own.imgh = c->sq_imgh;
            for (int k = 0; k < 3; k++) own.img[k] = c->sq_img[k];
            own.qpart = c->sq_qpart; own.qparta = c->sq_qparta; own.qparth = c->sq_qparth; own.stat = c->sq_stat;
            own.pk3 = c->sq_pk3; own.pk5 = c->sq_pk5; own.pk1 = c->sq_pk1;
            for (uint64_t t0 = 0; t0 < T; t0 += CH) {
                const int n = (int)(T - t0 < CH ? T - t0 : CH);
                if (mode == RWKV_MODE_PARRALEL) rc = enqueue_chunk(c, tokens + t0, n, t0, true);
                else rc = enqueue_pass(c, tokens + t0, n, t0, 0, ChunkPart{c->l0, c->l1, c->stream, &own});
                if (rc) return rc;
            }
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    for (uint64_t t = 0; t < T; t++) {
        c->h_ctl[t].token = tokens[t];
        c->h_ctl[t].slot = (mode == RWKV_MODE_PARRALEL) ? (unsigned)t : 0u;   // rwkv.cu:236-240
        c->h_ctl[t].out_row = (unsigned)t;
        c->h_ctl[t].step = 0; c->h_ctl[t].pad = 0;
        HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[t], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
        int rc = run_token(c, false);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

int rwkv_stage_forward(rwkv_ctx *c, uint64_t token, uint32_t slot, uint64_t *pick)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (slot >= c->maxT) return fail(RWKV_E_ARG, "state slot %u out of range (max context %llu)", slot, (unsigned long long)c->maxT);
    if (c->l0 == 0 && token >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id out of range");
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    c->h_ctl[0].token = token; c->h_ctl[0].slot = slot; c->h_ctl[0].out_row = slot; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    const bool last = c->l1 == c->L;
    int rc = run_token(c, last && pick != nullptr);
    if (rc) return rc;
    if (last && pick) HIPCHK(hipMemcpyAsync(pick, c->gen, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

double *rwkv_x_device(rwkv_ctx *c) { return c ? c->x : nullptr; }

int rwkv_set_state(rwkv_ctx *c, const double *xy, const double *aa, const double *bb, const double *pp,
                   const double *dd, uint64_t n_slots)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (n_slots > c->maxT) return fail(RWKV_E_ARG, "n_slots %llu > max context %llu", (unsigned long long)n_slots, (unsigned long long)c->maxT);
    const double *h[5] = {xy, aa, bb, pp, dd};
    const size_t bytes = n_slots * c->L * c->D * sizeof(double);
    HIPCHK(hipSetDevice(c->device));
    for (int s = 0; s < 5; s++)
        if (h[s]) HIPCHK(hipMemcpyAsync(c->state[s], h[s], bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

int rwkv_get_output(rwkv_ctx *c, float *logits, double *xy, double *aa, double *bb, double *pp, double *dd,
                    uint64_t T)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (T > c->maxT) return fail(RWKV_E_ARG, "n_tokens %llu > max context %llu", (unsigned long long)T, (unsigned long long)c->maxT);
    double *h[5] = {xy, aa, bb, pp, dd};
    HIPCHK(hipSetDevice(c->device));
    if (logits) HIPCHK(hipMemcpyAsync(logits, c->logits, T * RWKV_VOCAB * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    const size_t bytes = T * c->L * c->D * sizeof(double);
    for (int s = 0; s < 5; s++)
        if (h[s]) HIPCHK(hipMemcpyAsync(h[s], c->state[s], bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

int rwkv_reset_state(rwkv_ctx *c)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    HIPCHK(hipSetDevice(c->device));
    for (int s = 0; s < 5; s++)
        HIPCHK(hipMemsetAsync(c->state[s], 0, c->maxT * c->L * c->D * sizeof(double), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

int rwkv_decode_greedy(rwkv_ctx *c, uint64_t first_token, uint64_t n, uint64_t *out_tokens)
{
    if (!c || !out_tokens) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (c->l0 != 0 || c->l1 != c->L) return fail(RWKV_E_STATE, "needs a whole-model context (a pipeline stage has no embedding / head of its own)");
    if (first_token >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id out of range");
    if (n == 0 || n > c->gen_cap) return fail(RWKV_E_ARG, "n_tokens must be in 1..%u", c->gen_cap);
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    c->h_ctl[0].token = first_token; c->h_ctl[0].slot = 0; c->h_ctl[0].out_row = 0; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    for (uint64_t i = 0; i < n; i++) {
        int rc = run_token(c, true);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(out_tokens, c->gen, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

namespace {
int launch_typical(rwkv_ctx *c, int row, float temp, float tau, double u, uint64_t seed, bool use_seed, int flags, bool feedback)
{
    const bool ban0 = (flags & RWKV_SAMPLE_BAN0) != 0;
    TypicalArgs a;
    // default: what typical.h computes -- no cut (typical.h:50 assigns into a temporary), integer exponent
    // uint8(1/temp) (nc::power, typical.h:52); RWKV_SAMPLE_RECIPE: the documented recipe
    a.recipe = (flags & RWKV_SAMPLE_RECIPE) ? 1 : 0;
    if (a.recipe) a.expo = temp == 1.0f ? 1.0 : 1.0 / (double)temp;
    else { const double e = 1.0 / (double)temp; a.expo = temp == 1.0f ? 1.0 : (e >= 255.0 ? 255.0 : (double)(unsigned char)e); }
    a.logits = c->logits; a.row = row; a.ctl = c->ctl; a.gen = c->gen; a.gen_cap = c->gen_cap;
    a.temp = temp; a.tau = tau; a.u = u; a.seed = seed; a.use_seed = use_seed ? 1 : 0; a.ban0 = ban0 ? 1 : 0;
    a.feedback = feedback ? 1 : 0; a.pick = c->pick; a.part = c->ts_part; a.p = c->ts_p; a.pw = c->ts_pw; a.key = c->ts_key;
    k_typical_stats<<<dim3(TS_G), dim3(TS_GT), 0, c->stream>>>(a);
    k_typical_keys<<<dim3(TS_G), dim3(TS_GT), 0, c->stream>>>(a);
    k_typical<<<dim3(1), dim3(TS_NT), 0, c->stream>>>(a);
    HIPCHK(hipGetLastError());
    return 0;
}
} // namespace

int rwkv_sample_typical(rwkv_ctx *c, uint64_t row, float temp, float tau, double u, int flags, uint64_t *token)
{
    if (!c || !token) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (c->l1 != c->L) return fail(RWKV_E_STATE, "this pipeline stage does not hold the head");
    if (row >= c->maxT) return fail(RWKV_E_ARG, "logits row %llu out of range (max context %llu)", (unsigned long long)row, (unsigned long long)c->maxT);
    if (!(temp > 0.f) || !(u >= 0.0 && u < 1.0)) return fail(RWKV_E_ARG, "need temp > 0 and 0 <= u < 1");
    HIPCHK(hipSetDevice(c->device));
    int rc = launch_typical(c, (int)row, temp, tau, u, 0, false, flags, false);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(token, c->pick, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

int rwkv_decode_typical(rwkv_ctx *c, uint64_t first_token, uint64_t n, float temp, float tau, uint64_t seed, int flags, uint64_t *out_tokens)
{
    if (!c || !out_tokens) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (c->l0 != 0 || c->l1 != c->L) return fail(RWKV_E_STATE, "needs a whole-model context");
    if (first_token >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id out of range");
    if (n == 0 || n > c->gen_cap) return fail(RWKV_E_ARG, "n_tokens must be in 1..%u", c->gen_cap);
    if (!(temp > 0.f)) return fail(RWKV_E_ARG, "need temp > 0");
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    c->h_ctl[0].token = first_token; c->h_ctl[0].slot = 0; c->h_ctl[0].out_row = 0; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    for (uint64_t i = 0; i < n; i++) {
        int rc = run_token(c, false);                                   // the token graph without the argmax node
        if (!rc) rc = launch_typical(c, -1, temp, tau, 0.0, seed, true, flags | RWKV_SAMPLE_BAN0, true);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(out_tokens, c->gen, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

void rwkv_free(rwkv_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->g_fwd) { (void)hipGraphExecDestroy(c->g_fwd); c->g_fwd = nullptr; }
    if (c->g_greedy) { (void)hipGraphExecDestroy(c->g_greedy); c->g_greedy = nullptr; }
    rwkv_pipe_free(c);        // (no graphs left to re-capture)
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->herr) (void)hipHostFree(c->herr);
    for (auto &e : c->xs_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->hop_ev) if (e) (void)hipEventDestroy(e);
    for (auto &row : c->sp_done) for (auto &e : row) if (e) (void)hipEventDestroy(e);
    if (c->sp_end) (void)hipEventDestroy(c->sp_end);
    for (int k = 1; k < rwkv_ctx::SPLIT_MAX; k++) { if (c->sp_stream[k]) (void)hipStreamDestroy(c->sp_stream[k]); delete c->sp_scratch[k]; }
    for (auto &kv : c->sq_graphs) (void)hipGraphExecDestroy(kv.second);
    c->sq_graphs.clear();
    if (c->h_sq_tokens) (void)hipHostFree(c->h_sq_tokens);
    for (int r = 0; r < SQ_RING; r++) if (c->sq_ev[r]) (void)hipEventDestroy(c->sq_ev[r]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

float *rwkv_logits_device(rwkv_ctx *c) { return c ? c->logits : nullptr; }
double *rwkv_state_device(rwkv_ctx *c, int which) { return (c && which >= 0 && which < 5) ? c->state[which] : nullptr; }
void *rwkv_stream(rwkv_ctx *c) { return c ? (void *)c->stream : nullptr; }
int rwkv_abi_version(void) { return RWKV_MI355X_ABI_VERSION; }
uint64_t rwkv_resident_bytes(const rwkv_ctx *c) { return c ? (uint64_t)c->alloc_bytes : 0; }
int rwkv_decode_form(const rwkv_ctx *c) { return c && c->loaded ? (c->tile > 0 ? c->tile : 0) : -1; }

uint64_t rwkv_bytes_per_token(const rwkv_ctx *c)
{
    if (!c) return 0;
    const uint64_t nl = (c->l1 == UINT64_MAX ? c->L : c->l1) - c->l0, D = c->D, V = RWKV_VOCAB;
    return 13 * nl * D * D + (c->l1 == c->L || c->l1 == UINT64_MAX ? V * D : 0) + 168 * nl * D + 40 * D;   // SURVEY.md section 8(d), this stage's share
}

int rwkv_profile_token(rwkv_ctx *c, uint64_t token, int reps, double *ms, uint64_t *bytes, uint32_t *launches)
{
    if (!c || !ms) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (token >= RWKV_VOCAB || reps <= 0) return fail(RWKV_E_ARG, "bad token / reps");
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    const uint64_t L = c->L, D = c->D, V = RWKV_VOCAB;
    const int nev = (int)(4 * (c->l1 - c->l0) + 4);
    std::vector<hipEvent_t> ev(nev);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int k = 0; k < RWKV_N_KCLASS; k++) ms[k] = 0.0;
    c->h_ctl[0].token = token; c->h_ctl[0].slot = 0; c->h_ctl[0].out_row = 0; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    int rc = 0;
    for (int rep = 0; rep < reps && !rc; rep++) {
        rc = enqueue_token(c, true, ev.data());
        if (rc) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "sync failed"); break; }
        for (int i = 0; i + 1 < nev; i++) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "hipEventElapsedTime failed"); break; }
            const int nlay = (int)(c->l1 - c->l0);
            int cls;
            if (i == 0) cls = 0;
            else if (i <= 4 * nlay) cls = 1 + (i - 1) % 4;
            else if (i == 4 * nlay + 1) cls = 5;
            else cls = 6;
            ms[cls] += (double)t;
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (bytes) {
        bytes[0] = 4 * D; bytes[1] = 3 * D * D; bytes[2] = D * D; bytes[3] = 5 * D * D; bytes[4] = 4 * D * D;
        bytes[5] = V * D; bytes[6] = 0;
    }
    if (launches) {
        launches[0] = 1; launches[1] = launches[2] = launches[3] = launches[4] = (uint32_t)(c->l1 - c->l0); launches[5] = 1; launches[6] = 1;
    }
    return rc;
}

// Per-class launch duration from ONE event pair around a batch of back-to-back launches of that
// class (all L layers x reps): amortises the ~2.5 us a hipEvent pair adds per bracket, so the
// figure is comparable with rocprofv3's kernel durations.  ms[c] = total ms, n[c] = launches.
int rwkv_profile_batched(rwkv_ctx *c, uint64_t token, int reps, double *ms, uint32_t *n)
{
    if (!c || !ms || !n) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (token >= RWKV_VOCAB || reps <= 0) return fail(RWKV_E_ARG, "bad token / reps");
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    c->h_ctl[0].token = token; c->h_ctl[0].slot = 0; c->h_ctl[0].out_row = 0; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    int rc = enqueue_token(c, true, nullptr);   // valid inputs for every class
    if (rc) return rc;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    for (int cls = 0; cls < RWKV_N_KCLASS; cls++) {
        const bool per_layer = cls >= 1 && cls <= 4;
        // a class with ONE launch per token gets 8 x as many repetitions (4 launches in a bracket measured +10 % on k_head) and a
        // warm-up launch.  k_head's 206 MB would stay in the 256 MiB Infinity Cache between back-to-back launches (33.6 us against
        // 35.7 us inside the real token, rocprofv3): every head launch is preceded by ffn_v launches of four rotating layers (what a
        // token puts in front of it) and the same sequence WITHOUT the head is timed as well; the difference is the head.
        const int nrep = per_layer ? reps : reps * 8;
        const uint64_t nl = c->l1 - c->l0;
        const bool flush = cls == 5 && nl >= 4;
        float t = 0.f, t0 = 0.f;
        uint32_t cnt = 0;
        for (int pass = flush ? 0 : 1; pass < 2; pass++) {      // pass 0: the flushing launches alone
            launch_class(c, cls, per_layer ? c->l0 : 0);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipEventRecord(a, c->stream));
            cnt = 0;
            for (int r = 0; r < nrep; r++)
                for (uint64_t l = (per_layer ? c->l0 : 0); l < (per_layer ? c->l1 : 1); l++) {
                    if (flush) for (int q = 0; q < 4; q++) launch_class(c, 4, c->l0 + (uint64_t)(4 * r + q) % nl);
                    if (pass == 1) launch_class(c, cls, l);
                    cnt++;
                }
            HIPCHK(hipEventRecord(b, c->stream));
            HIPCHK(hipEventSynchronize(b));
            HIPCHK(hipEventElapsedTime(pass == 0 ? &t0 : &t, a, b));
        }
        ms[cls] = (double)(t - t0); n[cls] = cnt;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    HIPCHK(hipGetLastError());
    // the batches above advanced the recurrent state with a scrambled schedule: reset it
    for (int st = 0; st < 5; st++)
        HIPCHK(hipMemsetAsync(c->state[st], 0, c->maxT * c->L * c->D * sizeof(double), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

// debug: run one eager token with the phase timeline of the middle layer's ffn_rk kernel enabled;
// out receives grid*NW*8 100-MHz wall-clock stamps (see tl_stamp in kernels.hip.h)
int rwkv_debug_timeline(rwkv_ctx *c, uint64_t token, unsigned long long *out, uint64_t cap)
{
    if (!c || !out) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    const size_t n = (size_t)(c->grid > 512 ? c->grid : 512) * NW * 8;      // (a chunk GEMM may launch more workgroups than the decode grid)
    if (cap < n) return fail(RWKV_E_ARG, "need room for %zu stamps", n);
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    if (!c->tl) { int rc = dalloc(c, &c->tl, n); if (rc) return rc; }
    HIPCHK(hipMemsetAsync(c->tl, 0, n * 8, c->stream));
    c->h_ctl[0].token = token; c->h_ctl[0].slot = 0; c->h_ctl[0].out_row = 0; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    { const char *e = getenv("RWKV_TL_CLASS"); c->tl_cls = e ? atoi(e) : 3; }
    c->tl_on = true;
    int rc;
    if (c->tl_cls >= 10) {          // a GEMM of the chunk path (10 + kind: one 32-row GPT chunk of this token; 20 + kind: a 64-row pass, the two-half GEMMs)
        if (!c->seq_ok) { c->tl_on = false; return fail(RWKV_E_STATE, "the chunk path is not loaded (max_ctx 1)"); }
        uint64_t toks[SEQ_TM];
        for (int t = 0; t < SEQ_TM; t++) toks[t] = token;
        int rows = SEQ_T;
        if (c->tl_cls >= 20) { c->tl_cls -= 10; if (c->seq_rows > SEQ_T && c->maxT >= (uint64_t)SEQ_TM) rows = SEQ_TM; }
        rc = enqueue_chunk(c, toks, rows, 0, false);
    } else rc = enqueue_token(c, false, nullptr);
    c->tl_on = false;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, c->tl, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

// ---- per-kernel parity hooks (tests/test_kernels_gpu.py): the PRODUCTION launch of one decode kernel class, eagerly, on whatever the
// context's buffers hold, and access to the intermediate vectors the kernels hand to each other.  A token is rwkv_debug_launch(0),
// then (1, 2, 3, 4) per layer, then 5: exactly what enqueue_token launches, one kernel at a time. ----
int rwkv_debug_launch(rwkv_ctx *c, int cls, uint64_t layer, uint64_t token, uint32_t slot)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (cls < 0 || cls > 6) return fail(RWKV_E_ARG, "kernel class %d out of range (0 first, 1 att, 2 att_out, 3 ffn r+k, 4 ffn_v, 5 head, 6 argmax)", cls);
    if (cls >= 1 && cls <= 4 && (layer < c->l0 || layer >= c->l1)) return fail(RWKV_E_ARG, "layer %llu is not one of this context's", (unsigned long long)layer);
    if (cls >= 5 && c->l1 != c->L) return fail(RWKV_E_STATE, "this context does not hold the head");
    if (slot >= c->maxT) return fail(RWKV_E_ARG, "state slot %u out of range", slot);
    if (c->l0 == 0 && token >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id out of range");
    HIPCHK(hipSetDevice(c->device));
    { const int rcp = begin_call(c); if (rcp) return rcp; }
    if (cls == 0) {      // the token's control block: every later launch of the token reads slot / out_row from it
        c->h_ctl[0].token = token; c->h_ctl[0].slot = slot; c->h_ctl[0].out_row = slot; c->h_ctl[0].step = 0; c->h_ctl[0].pad = 0;
        HIPCHK(hipMemcpyAsync(c->ctl, &c->h_ctl[0], sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    }
    launch_class(c, cls, layer);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}

namespace {
// device address and byte size of intermediate buffer `which` (enum rwkv_debug_buf)
int debug_buf(rwkv_ctx *c, int which, void **p, size_t *bytes)
{
    const size_t D = (size_t)c->D, G = (size_t)c->grid;
    switch (which) {
    case RWKV_DBG_X: *p = c->x; *bytes = D * 8; break;
    case RWKV_DBG_YBUF: *p = c->ybuf; *bytes = D * 4; break;
    case RWKV_DBG_PART_ATT: *p = c->partA; *bytes = G * 8; break;
    case RWKV_DBG_PMAX_ATT: *p = c->partMA; *bytes = G * 4; break;
    case RWKV_DBG_HBUF: *p = c->hbuf; *bytes = 4 * D * 4; break;
    case RWKV_DBG_RGATE: *p = c->rgate; *bytes = D * 4; break;
    case RWKV_DBG_PART_FFN: *p = c->partF; *bytes = G * 8; break;
    case RWKV_DBG_PMAX_FFN: *p = c->partMF; *bytes = G * 4; break;
    case RWKV_DBG_LNSTAT: *p = c->lnstat; *bytes = 6 * 8; break;
    default: return fail(RWKV_E_ARG, "no such debug buffer: %d", which);
    }
    return 0;
}
} // namespace

int rwkv_debug_read(rwkv_ctx *c, int which, void *dst, uint64_t cap)
{
    if (!c || !dst) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    void *p = nullptr; size_t n = 0;
    { const int rc = debug_buf(c, which, &p, &n); if (rc) return rc; }
    if (cap < n) return fail(RWKV_E_ARG, "debug buffer %d holds %zu bytes", which, n);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int rwkv_debug_write(rwkv_ctx *c, int which, const void *src, uint64_t bytes)
{
    if (!c || !src) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    void *p = nullptr; size_t n = 0;
    { const int rc = debug_buf(c, which, &p, &n); if (rc) return rc; }
    if (bytes != n) return fail(RWKV_E_ARG, "debug buffer %d holds %zu bytes", which, n);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

uint64_t rwkv_debug_grid(const rwkv_ctx *c) { return c ? (uint64_t)c->grid : 0; }

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Layer pipeline, native transport.  No reference counterpart (the reference is single-device); the
// north_star asks for "layers optionally pipelined across the 8 GPUs of one node via RCCL send/recv over xGMI".
// The hop lives INSIDE the engine: ncclSend / ncclRecv are enqueued on the engine's own stream between the stage's
// graph launches, the greedy id travels last stage -> first stage as a device-to-device u64 straight into the control
// block the next launch reads, and the host never waits inside the loop (one synchronisation at the very end).
namespace {

int pipe_fail(Pipe *p, int rc, const char *what)
{
    return fail(RWKV_E_DEVICE, "%s: %s", what, (p && p->GetErrorString) ? p->GetErrorString(rc) : "RCCL error");
}
#define NCHK(expr) do { int r__ = (expr); if (r__ != 0) return pipe_fail(p, r__, #expr); } while (0)

int pipe_open(Pipe *p)
{
    if (p->lib) return 0;
    // Which RCCL: (1) RWKV_RCCL_LIB if set; (2) the copy the PROCESS HAS ALREADY LOADED -- a host that imported PyTorch-ROCm carries
    // torch/lib/librccl.so, built against the HIP runtime torch brought along, and the engine's streams and buffers live in that same
    // runtime: a second RCCL of another ROCm release beside it is a version mix nobody tests; (3) the system's, for a host without one.
    const char *env = getenv("RWKV_RCCL_LIB");
    if (env && env[0]) p->lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    if (!p->lib && !(env && env[0]))
        for (const char *n : {"librccl.so", "librccl.so.1"}) {
            p->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (p->lib) break;
        }
    if (!p->lib && !(env && env[0]))
        for (const char *n : {"librccl.so.1", "librccl.so"}) {
            p->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (p->lib) break;
        }
    if (!p->lib) return fail(RWKV_E_DEVICE, "cannot load %s (%s)", (env && env[0]) ? env : "librccl.so", dlerror());
    auto sym = [&](const char *n) { return dlsym(p->lib, n); };
    *(void **)&p->GetUniqueId = sym("ncclGetUniqueId");
    *(void **)&p->CommInitRank = sym("ncclCommInitRank");
    *(void **)&p->CommDestroy = sym("ncclCommDestroy");
    *(void **)&p->Send = sym("ncclSend");
    *(void **)&p->Recv = sym("ncclRecv");
    *(void **)&p->GroupStart = sym("ncclGroupStart");
    *(void **)&p->GroupEnd = sym("ncclGroupEnd");
    *(void **)&p->GetErrorString = sym("ncclGetErrorString");
    *(void **)&p->GetVersion = sym("ncclGetVersion");      // (optional: diagnostics only)
    if (!p->GetUniqueId || !p->CommInitRank || !p->Send || !p->Recv || !p->GroupStart || !p->GroupEnd)
        return fail(RWKV_E_DEVICE, "librccl.so lacks the point-to-point API");
    return 0;
}
constexpr int kNcclUint64 = 5, kNcclFloat64 = 8;

} // namespace

extern "C" {

// which RCCL the pipeline transport binds in this process (path of the shared object that holds ncclSend); loads it if need be.
// No GPU involved: a host can check the pairing before it builds a pipeline.
int rwkv_pipe_rccl_path(char *out, uint64_t cap)
{
    if (!out || cap == 0) return fail(RWKV_E_ARG, "NULL argument");
    static Pipe probe;
    int rc = pipe_open(&probe);
    if (rc) return rc;
    Dl_info info{};
    if (!dladdr(reinterpret_cast<void *>(probe.Send), &info) || !info.dli_fname) return fail(RWKV_E_DEVICE, "dladdr failed for ncclSend");
    snprintf(out, (size_t)cap, "%s", info.dli_fname);
    return 0;
}

// 128 opaque bytes (ncclUniqueId) made by ONE rank and handed to every rank of the pipeline out of band
int rwkv_pipe_unique_id(void *out128)
{
    if (!out128) return fail(RWKV_E_ARG, "NULL argument");
    static Pipe boot;
    Pipe *p = &boot;
    int rc = pipe_open(p);
    if (rc) return rc;
    Pipe::Id id;
    NCHK(p->GetUniqueId(&id));
    memcpy(out128, id.b, sizeof(id.b));
    return 0;
}

int rwkv_pipe_init(rwkv_ctx *c, const void *id128, int rank, int world)
{
    if (!c || !id128) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (world < 1 || rank < 0 || rank >= world) return fail(RWKV_E_ARG, "bad rank %d of %d", rank, world);
    if (c->pipe) return fail(RWKV_E_STATE, "pipeline transport already initialised");
    if ((rank == 0) != (c->l0 == 0) || (rank == world - 1) != (c->l1 == c->L))
        return fail(RWKV_E_ARG, "rank %d of %d does not match this context's layer range [%llu, %llu) of %llu", rank, world,
                    (unsigned long long)c->l0, (unsigned long long)c->l1, (unsigned long long)c->L);
    HIPCHK(hipSetDevice(c->device));
    Pipe *p = new Pipe();
    int rc = pipe_open(p);
    if (rc) { delete p; return rc; }
    Pipe::Id id;
    memcpy(id.b, id128, sizeof(id.b));
    int r = p->CommInitRank(&p->comm, world, id, rank);
    if (r != 0) { rc = pipe_fail(p, r, "ncclCommInitRank"); delete p; return rc; }
    p->rank = rank; p->world = world;
    // every failure from here on gives the communicator back and leaves the context as it was (hop read from c->x)
    auto undo = [&](int code) {
        if (p->CommDestroy) (void)p->CommDestroy(p->comm);
        delete p;
        c->x_in = nullptr;
        return code;
    };
    if (rank > 0) {
        if ((rc = dalloc(c, &c->x_in, c->D))) return undo(rc);
        // the stage graphs captured at load time read the hop from c->x: rebuild them around x_in
        if ((rc = rebuild_graphs(c))) {
            c->x_in = nullptr;
            (void)rebuild_graphs(c);          // back to the graphs that read c->x (on failure: no graphs, eager launches)
            return undo(rc);
        }
    }
    // Every rank must cut a prompt into the SAME micro-batches (rwkv_pipe_prefill: rows per ncclSend / ncclRecv) and speak about the same
    // model: the values behind that are per-rank (RWKV_SEQ_ROWS in each process's environment, the max_ctx each context was loaded with),
    // so they are agreed here, once, over the communicator itself -- a ring pass of (min, max) per field, world - 1 hops, after which every
    // rank holds the same extremes and every rank fails the same way on a mismatch (a pipeline that disagrees would hang in its first
    // prefill hop, or corrupt the residual stream).
    {
        const uint64_t my_ch = !c->seq_ok ? 0 : (c->seq_rows > SEQ_T && c->maxT >= (uint64_t)SEQ_TM) ? (uint64_t)SEQ_TM : (c->maxT >= (uint64_t)SEQ_T ? (uint64_t)SEQ_T : 0);
        uint64_t mm[8] = {my_ch, my_ch, c->D, c->D, c->L, c->L, (uint64_t)world, (uint64_t)world};      // (min, max) pairs
        {   // (world == 1: one hop to SELF -- nothing to agree on, but it is the one ncclSend / ncclRecv group a single process can make on the
            //  real library: entry points, datatype codes, the group and the stream order are exercised before any pipeline depends on them)
            uint64_t *d = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&d), 16 * sizeof(uint64_t)) != hipSuccess) return undo(fail(RWKV_E_DEVICE, "rwkv_pipe_init: no device memory for the agreement round"));
            int r = 0;
            hipError_t e = hipSuccess;
            bool self_ok = true;
            const int hops = world > 1 ? world - 1 : 1;
            for (int hop = 0; hop < hops && !r && e == hipSuccess; hop++) {
                e = hipMemcpyAsync(d, mm, sizeof(mm), hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipMemsetAsync(d + 8, 0xff, 8 * sizeof(uint64_t), c->stream);
                if (e != hipSuccess) break;
                r = p->GroupStart();
                if (!r) r = p->Send(d, 8, kNcclUint64, (rank + 1) % world, p->comm, c->stream);
                if (!r) r = p->Recv(d + 8, 8, kNcclUint64, (rank + world - 1) % world, p->comm, c->stream);
                const int r2 = p->GroupEnd();
                if (!r) r = r2;
                uint64_t got[8];
                if (!r) e = hipMemcpyAsync(got, d + 8, sizeof(got), hipMemcpyDeviceToHost, c->stream);
                if (!r && e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (!r && e == hipSuccess) {
                    if (world == 1) self_ok = memcmp(got, mm, sizeof(got)) == 0;
                    for (int k = 0; k < 8; k += 2) { mm[k] = std::min(mm[k], got[k]); mm[k + 1] = std::max(mm[k + 1], got[k + 1]); }
                }
            }
            (void)hipFree(d);
            if (r) { rc = pipe_fail(p, r, "agreement round of rwkv_pipe_init"); if (rank > 0) { c->x_in = nullptr; (void)rebuild_graphs(c); } return undo(rc); }
            if (e != hipSuccess) { rc = fail(RWKV_E_DEVICE, "agreement round of rwkv_pipe_init: %s", hipGetErrorString(e)); if (rank > 0) { c->x_in = nullptr; (void)rebuild_graphs(c); } return undo(rc); }
            if (!self_ok) return undo(fail(RWKV_E_DEVICE, "rwkv_pipe_init: a grouped ncclSend / ncclRecv of 8 x uint64 to this rank itself did not deliver what was sent"));
        }
        if (mm[0] != mm[1] || mm[2] != mm[3] || mm[4] != mm[5] || mm[6] != mm[7]) {
            rc = fail(RWKV_E_ARG, "the ranks of this pipeline disagree: prefill micro-batch rows %llu..%llu (RWKV_SEQ_ROWS / max_ctx differ between ranks; this rank: %llu), "
                                  "n_embed %llu..%llu, n_layers %llu..%llu, world %llu..%llu", (unsigned long long)mm[0], (unsigned long long)mm[1], (unsigned long long)my_ch,
                      (unsigned long long)mm[2], (unsigned long long)mm[3], (unsigned long long)mm[4], (unsigned long long)mm[5], (unsigned long long)mm[6], (unsigned long long)mm[7]);
            if (rank > 0) { c->x_in = nullptr; (void)rebuild_graphs(c); }
            return undo(rc);
        }
        p->ch = my_ch;
    }
    {   // what this rank's end of the transport is made of: for the first run on real xGMI to be diagnosable from its log / bench line
        int ver = -1;
        if (p->GetVersion) (void)p->GetVersion(&ver);
        Dl_info di{};
        const char *path = (dladdr(reinterpret_cast<void *>(p->Send), &di) && di.dli_fname) ? di.dli_fname : "?";
        char bus[64] = "?";
        (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->device);
        hipDeviceProp_t prop{};
        (void)hipGetDeviceProperties(&prop, c->device);
        int rt = 0;
        (void)hipRuntimeGetVersion(&rt);
        char buf[1024];
        snprintf(buf, sizeof(buf), "{\"rank\": %d, \"world\": %d, \"layers\": [%llu, %llu], \"device\": %d, \"pci_bus_id\": \"%s\", \"arch\": \"%s\", \"cus\": %d, "
                                   "\"rccl_version\": %d, \"rccl_path\": \"%s\", \"hip_runtime\": %d, \"prefill_rows\": %llu, \"HSA_ENABLE_IPC_MODE_LEGACY\": \"%s\"}",
                 rank, world, (unsigned long long)c->l0, (unsigned long long)c->l1, c->device, bus, prop.gcnArchName, prop.multiProcessorCount, ver, path, rt,
                 (unsigned long long)p->ch, getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "");
        p->info = buf;
        fprintf(stderr, "[rwkv_mi355x] pipeline transport up: %s\n", buf);
    }
    c->pipe = p;
    return 0;
}

// this rank's end of the pipeline transport as one JSON object (rank, world, layer range, device ordinal, PCI bus id, arch, RCCL
// version code and the path of the shared object that holds ncclSend, HIP runtime version, agreed prefill micro-batch rows)
int rwkv_pipe_info(rwkv_ctx *c, char *out, uint64_t cap)
{
    if (!c || !out || cap == 0) return fail(RWKV_E_ARG, "NULL argument");
    if (!c->pipe) return fail(RWKV_E_STATE, "needs rwkv_pipe_init done");
    snprintf(out, (size_t)cap, "%s", c->pipe->info.c_str());
    return 0;
}

// Greedy decode of `world` independent streams (stream k starts from first_tokens[k], state slot k), n_steps tokens each,
// with the model's layers pipelined over the ranks: at tick t rank r works on item j = t - r (stream j % S, step j / S),
// so every GPU is busy once the pipe is full.  Per tick and rank, all on the engine stream:
//   control block of the item (pinned ring -> device)   -> ONE RCCL group { send x to r+1 | send the previous pick to
//   rank 0 (last rank) | recv x from r-1 | recv the fed-back id into the control block (rank 0) }   -> the stage's graph.
// first_tokens: [world] (read on rank 0).  picks: [world][n_steps] (written on the last rank; may be NULL elsewhere).
// n_streams < world leaves the other streams' slots of the schedule empty: n_streams = 1 is ONE stream through all the stages, i.e.
// the latency of single-stream decode on the pipeline, t_tok + (S - 1) hops + the fed-back id (SURVEY 8e "expected scaling").
// With rwkv_pipe_profile(ctx, 1) every tick whose RCCL group holds a receive is bracketed by an event pair (rwkv_pipe_hop_stats).
int rwkv_pipe_decode_streams(rwkv_ctx *c, const uint64_t *first_tokens, uint64_t n_steps, uint64_t n_streams, uint64_t *picks)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded || !c->pipe) return fail(RWKV_E_STATE, "needs a loaded context with rwkv_pipe_init done");
    Pipe *p = c->pipe;
    const int S = p->world, rank = p->rank;
    const bool lastr = rank == S - 1;
    const uint64_t n_items = (uint64_t)S * n_steps;
    if (n_streams == 0 || n_streams > (uint64_t)S) return fail(RWKV_E_ARG, "n_streams must be in 1..world");
    if (n_steps == 0 || n_items > c->gen_cap) return fail(RWKV_E_ARG, "world * n_steps must be in 1..%u", c->gen_cap);
    if ((uint64_t)S > c->maxT) return fail(RWKV_E_ARG, "needs max_ctx >= %d state slots (one per stream in flight)", S);
    if (rank == 0 && !first_tokens) return fail(RWKV_E_ARG, "rank 0 needs first_tokens");
    if (lastr && !picks) return fail(RWKV_E_ARG, "the last rank needs picks");
    // A bad id must not strand the other ranks in their ncclRecv: rank 0 runs the schedule with id 0 in its place and reports
    // the error when the schedule has drained (the argument checks above depend only on values every rank shares).
    bool bad_id = false;
    if (rank == 0)
        for (int k = 0; k < (int)n_streams; k++) bad_id = bad_id || first_tokens[k] >= RWKV_VOCAB;
    HIPCHK(hipSetDevice(c->device));
    // (a failure here must not strand the peers in their ncclRecv either: it is reported once the schedule has drained; without
    // re-captured graphs run_token launches the kernels one by one)
    const int rc_begin = begin_call(c);
    if (c->pipe_ring_cap < n_items) {
        if (c->pipe_ring) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->pipe_ring); c->pipe_ring = nullptr; c->pipe_ring_cap = 0; }
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->pipe_ring), sizeof(Ctl) * n_items, hipHostMallocDefault));
        c->pipe_ring_cap = n_items;
    }
    Ctl *ring = c->pipe_ring;
    for (uint64_t j = 0; j < n_items; j++) {
        const uint64_t stream = j % S, step = j / S;
        ring[j].token = (rank == 0 && step == 0 && stream < n_streams) ? (first_tokens[stream] < RWKV_VOCAB ? first_tokens[stream] : 0) : 0;
        ring[j].slot = (unsigned)stream; ring[j].out_row = (unsigned)stream; ring[j].step = (unsigned)j; ring[j].pad = 0;
    }
    auto active = [&](uint64_t j) { return j % (uint64_t)S < n_streams; };
    auto has_work = [&](int r, uint64_t t) { return t >= (uint64_t)r && t - r < n_items && active(t - r); };
    int rc = 0;
    size_t hop_n = 0;
    for (uint64_t tick = 0; tick < n_items + S - 1 && !rc; tick++) {
        const bool work = has_work(rank, tick);
        const bool feedback = tick >= (uint64_t)S && tick < n_items && active(tick);   // stage 0 starts a step >= 1: its token is the last stage's pick
        const uint64_t j = tick - rank;
        if (work && hipMemcpyAsync(c->ctl, &ring[j], sizeof(Ctl), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "control block copy failed"); break; }
        bool timed = c->pipe_prof && S > 1 && ((rank > 0 && work) || (rank == 0 && feedback)) && hop_n < rwkv_ctx::HOP_EV;
        if (timed) {      // (a failure here only drops the measurement: the schedule must go on, the peers are waiting)
            if (!c->hop_ev[2 * hop_n] && (hipEventCreate(&c->hop_ev[2 * hop_n]) != hipSuccess || hipEventCreate(&c->hop_ev[2 * hop_n + 1]) != hipSuccess)) timed = false;
            if (timed && hipEventRecord(c->hop_ev[2 * hop_n], c->stream) != hipSuccess) timed = false;
        }
        if (S > 1) {
            int r = p->GroupStart();
            if (!r && rank < S - 1 && has_work(rank + 1, tick)) r = p->Send(c->x, c->D, kNcclFloat64, rank + 1, p->comm, c->stream);
            if (!r && lastr && feedback) r = p->Send(c->gen + (tick - S), 1, kNcclUint64, 0, p->comm, c->stream);   // item tick - S: produced here one tick ago
            if (!r && rank > 0 && work) r = p->Recv(c->x_in, c->D, kNcclFloat64, rank - 1, p->comm, c->stream);
            if (!r && rank == 0 && feedback) r = p->Recv(&c->ctl->token, 1, kNcclUint64, S - 1, p->comm, c->stream);
            const int r2 = p->GroupEnd();
            if (r || r2) { rc = pipe_fail(p, r ? r : r2, "RCCL hop"); break; }
            if (timed && hipEventRecord(c->hop_ev[2 * hop_n + 1], c->stream) == hipSuccess) hop_n++;
        } else if (feedback) {
            // one stage: the pick of the previous step is already in ctl->token ... but the control block was just overwritten
            HIPCHK(hipMemcpyAsync(&c->ctl->token, c->gen + (j - 1), sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        }
        if (work) rc = run_token(c, lastr);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e != hipSuccess) rc = fail(RWKV_E_DEVICE, "pipeline decode: %s", hipGetErrorString(e));
    if (!rc && lastr) {
        std::vector<uint64_t> g(n_items);
        if (hipMemcpy(g.data(), c->gen, n_items * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RWKV_E_DEVICE, "copy of the picks failed");
        else for (uint64_t j = 0; j < n_items; j++) picks[(j % S) * n_steps + j / S] = active(j) ? g[j] : 0;     // (rows of streams that did not run: zero)
    }
    { const int dc = device_check(c); if (!rc) rc = dc; }      // always consumed: a code raised here must not surface from a later call
    if (!rc && rc_begin) rc = rc_begin;
    if (!rc && c->pipe_prof) {
        double sum = 0.0, mn = 1e30, mx = 0.0;
        for (size_t i = 0; i < hop_n; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->hop_ev[2 * i], c->hop_ev[2 * i + 1]) != hipSuccess) continue;
            sum += ms; mn = std::min(mn, (double)ms); mx = std::max(mx, (double)ms);
        }
        c->hop_stats[0] = (double)hop_n; c->hop_stats[1] = hop_n ? 1e3 * sum / hop_n : 0.0; c->hop_stats[2] = hop_n ? 1e3 * mn : 0.0; c->hop_stats[3] = 1e3 * mx;
    }
    if (!rc && bad_id) rc = fail(RWKV_E_ARG, "token id out of range (the schedule ran with id 0 in its place)");
    return rc;
}

// ---- 2 x world streams in flight on TWO communicators: the hop under the compute -------------------------------------------------------
// rwkv_pipe_decode* above runs hop and stage strictly one after the other on the engine's stream: every tick costs t_hop + t_stage.  Here a
// rank keeps TWO independent copies of that schedule going -- the streams of even index on communicator 0 and HIP stream cs[0], those of odd
// index on communicator 1 and cs[1], each with its own hop buffers -- and ONE compute stream (the engine's) that alternates between them:
// while the stage works on parity 0's item, parity 1's RCCL group {send the previous result | recv the next input} is in flight, and vice
// versa.  A tick costs max(t_stage, t_hop) once both are primed.  Two communicators, because operations of one communicator are serialised
// in the order they were enqueued whatever stream they are on: a receive that waits for its peer would hold the other parity's send behind
// it.  Each sub-schedule is the tick schedule of rwkv_pipe_decode_streams (same group composition, same deadlock argument); events order
// the comm streams against the compute stream: ev_rx[k] (the group's receives have landed) and ev_cp[k] (the stage's output is in
// xout[k], xin[k] has been consumed).  Results are those of 2 x world independent greedy decodes (stream g on state slot g).
namespace {
// streams, hop buffers and events of the two-communicator schedule (not the communicator itself): idempotent, also the clean-up of a set-up
// that failed half way
void pipe_dual_teardown(Pipe *p)
{
    for (int k = 0; k < 2; k++) {
        if (p->cs[k]) { (void)hipStreamSynchronize(p->cs[k]); (void)hipStreamDestroy(p->cs[k]); p->cs[k] = nullptr; }
        if (p->xin[k]) { (void)hipFree(p->xin[k]); p->xin[k] = nullptr; }
        if (p->xout[k]) { (void)hipFree(p->xout[k]); p->xout[k] = nullptr; }
        if (p->idbuf[k]) { (void)hipFree(p->idbuf[k]); p->idbuf[k] = nullptr; }
        if (p->ev_rx[k]) { (void)hipEventDestroy(p->ev_rx[k]); p->ev_rx[k] = nullptr; }
        if (p->ev_cp[k]) { (void)hipEventDestroy(p->ev_cp[k]); p->ev_cp[k] = nullptr; }
    }
    p->dual_ready = false;
}
// Re-entrant: `dual_ready` is set only when every resource AND the second communicator exist; a call that fails half way
// frees what it made and the next call starts over.  The id exchange always completes on every rank: if rank 0 cannot make an id it sends
// an all-zero one, which every rank (rank 0 included) treats as the failure it is -- nobody is left blocked in a receive.
int pipe_dual_setup(rwkv_ctx *c)
{
    Pipe *p = c->pipe;
    if (p->dual_ready) return 0;
    pipe_dual_teardown(p);
    const int S = p->world, rank = p->rank;
    if (S > 1 && !p->comm2) {
        // the second communicator's id is made by rank 0 and travels over the first one (16 x u64)
        unsigned long long *d = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(Pipe::Id)));
        Pipe::Id id;
        memset(&id, 0, sizeof(id));
        int r = 0, r_id = 0;
        if (rank == 0) {
            r_id = p->GetUniqueId(&id);
            if (r_id) memset(&id, 0, sizeof(id));               // the peers learn of the failure from the id itself
            if (hipMemcpyAsync(d, id.b, sizeof(id.b), hipMemcpyHostToDevice, c->stream) != hipSuccess) r = -1;
            if (!r) {
                r = p->GroupStart();
                for (int q = 1; q < S && !r; q++) r = p->Send(d, sizeof(id.b) / 8, kNcclUint64, q, p->comm, c->stream);
                const int r2 = p->GroupEnd();
                if (!r) r = r2;
            }
        } else {
            r = p->Recv(d, sizeof(id.b) / 8, kNcclUint64, 0, p->comm, c->stream);
            if (!r && hipMemcpyAsync(id.b, d, sizeof(id.b), hipMemcpyDeviceToHost, c->stream) != hipSuccess) r = -1;
        }
        const hipError_t e = hipStreamSynchronize(c->stream);
        (void)hipFree(d);
        if (r_id) return pipe_fail(p, r_id, "ncclGetUniqueId (second communicator)");
        if (r) return pipe_fail(p, r > 0 ? r : 1, "exchange of the second communicator's id");
        HIPCHK(e);
        bool zero = true;
        for (size_t q = 0; q < sizeof(id.b); q++) zero = zero && id.b[q] == 0;
        if (zero) return fail(RWKV_E_DEVICE, "rank 0 could not make the second communicator's id (it sent the all-zero id)");
        r = p->CommInitRank(&p->comm2, S, id, rank);
        if (r) { p->comm2 = nullptr; return pipe_fail(p, r, "ncclCommInitRank (second communicator)"); }
    }
    auto make = [&]() -> int {
        for (int k = 0; k < 2; k++) {
            HIPCHK(hipStreamCreateWithFlags(&p->cs[k], hipStreamNonBlocking));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&p->xin[k]), c->D * sizeof(double)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&p->xout[k]), c->D * sizeof(double)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&p->idbuf[k]), 64));
            HIPCHK(hipEventCreateWithFlags(&p->ev_rx[k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_cp[k], hipEventDisableTiming));
        }
        return 0;
    };
    const int rm = make();
    if (rm) { pipe_dual_teardown(p); return rm; }
    p->dual_ready = true;
    return 0;
}
} // namespace

// first_tokens: [2 * world] (read on rank 0); picks: [2 * world][n_steps] (written on the last rank).  Needs max_ctx >= 2 * world state slots.
int rwkv_pipe_decode_dual(rwkv_ctx *c, const uint64_t *first_tokens, uint64_t n_steps, uint64_t *picks)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded || !c->pipe) return fail(RWKV_E_STATE, "needs a loaded context with rwkv_pipe_init done");
    Pipe *p = c->pipe;
    const int S = p->world, rank = p->rank;
    const bool lastr = rank == S - 1;
    const uint64_t n_sub = (uint64_t)S * n_steps;             // items of ONE sub-schedule
    if (n_steps == 0 || 2 * n_sub > c->gen_cap) return fail(RWKV_E_ARG, "2 * world * n_steps must be in 1..%u", c->gen_cap);
    if ((uint64_t)(2 * S) > c->maxT) return fail(RWKV_E_ARG, "needs max_ctx >= %d state slots (two streams per stage in flight)", 2 * S);
    if (rank == 0 && !first_tokens) return fail(RWKV_E_ARG, "rank 0 needs first_tokens");
    if (lastr && !picks) return fail(RWKV_E_ARG, "the last rank needs picks");
    bool bad_id = false;
    if (rank == 0)
        for (int g = 0; g < 2 * S; g++) bad_id = bad_id || first_tokens[g] >= RWKV_VOCAB;
    HIPCHK(hipSetDevice(c->device));
    const int rc_begin = begin_call(c);
    { const int rs = pipe_dual_setup(c); if (rs) return rs; }
    if (c->pipe_ring_cap < 2 * n_sub) {
        if (c->pipe_ring) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->pipe_ring); c->pipe_ring = nullptr; c->pipe_ring_cap = 0; }
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->pipe_ring), sizeof(Ctl) * 2 * n_sub, hipHostMallocDefault));
        c->pipe_ring_cap = 2 * n_sub;
    }
    // item (k, i): sub-schedule k, sub-item i -> global stream g = 2 (i % S) + k, step i / S, id J = 2 i + k (its slot in the pick list)
    Ctl *ring = c->pipe_ring;
    for (uint64_t i = 0; i < n_sub; i++)
        for (int k = 0; k < 2; k++) {
            const uint64_t J = 2 * i + k, g = 2 * (i % S) + k, step = i / S;
            ring[J].token = (rank == 0 && step == 0) ? (first_tokens[g] < RWKV_VOCAB ? first_tokens[g] : 0) : 0;
            ring[J].slot = (unsigned)g; ring[J].out_row = (unsigned)g; ring[J].step = (unsigned)J; ring[J].pad = 0;
        }
    auto has_work = [&](int r, uint64_t t) { return t >= (uint64_t)r && t - r < n_sub; };
    int rc = 0;
    // everything enqueued so far on the engine's stream precedes the comm streams' first operation
    for (int k = 0; k < 2 && !rc; k++) {
        if (hipEventRecord(p->ev_cp[k], c->stream) != hipSuccess || hipStreamWaitEvent(p->cs[k], p->ev_cp[k], 0) != hipSuccess) rc = fail(RWKV_E_DEVICE, "event setup failed");
    }
    for (uint64_t tick = 0; tick < n_sub + S - 1 && !rc; tick++) {
        const bool work = has_work(rank, tick);
        const bool feedback = tick >= (uint64_t)S && tick < n_sub;       // stage 0 starts a step >= 1: its token is the last stage's pick
        const uint64_t i = tick - rank;
        // ---- the hops of both parities, each on its own communicator and stream ----
        for (int k = 0; k < 2 && !rc && S > 1; k++) {
            void *comm = k == 0 ? p->comm : p->comm2;
            hipStream_t cs = p->cs[k];
            if (tick > 0 && has_work(rank, tick - 1) && hipStreamWaitEvent(cs, p->ev_cp[k], 0) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "event wait failed"); break; }
            int r = p->GroupStart();
            if (!r && rank < S - 1 && has_work(rank + 1, tick)) r = p->Send(p->xout[k], c->D, kNcclFloat64, rank + 1, comm, cs);
            if (!r && lastr && feedback) r = p->Send(c->gen + (2 * (tick - S) + k), 1, kNcclUint64, 0, comm, cs);      // sub-item tick - S: finished here one tick ago
            if (!r && rank > 0 && work) r = p->Recv(p->xin[k], c->D, kNcclFloat64, rank - 1, comm, cs);
            if (!r && rank == 0 && feedback) r = p->Recv(p->idbuf[k], 1, kNcclUint64, S - 1, comm, cs);
            const int r2 = p->GroupEnd();
            if (r || r2) { rc = pipe_fail(p, r ? r : r2, "RCCL hop"); break; }
            if (hipEventRecord(p->ev_rx[k], cs) != hipSuccess) rc = fail(RWKV_E_DEVICE, "event record failed");
        }
        // ---- the stage, parity 0 then parity 1, on the engine's stream ----
        for (int k = 0; k < 2 && !rc && work; k++) {
            const uint64_t J = 2 * i + k;
            if (S > 1 && hipStreamWaitEvent(c->stream, p->ev_rx[k], 0) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "event wait failed"); break; }
            if (hipMemcpyAsync(c->ctl, &ring[J], sizeof(Ctl), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "control block copy failed"); break; }
            if (feedback && rank == 0) {
                const void *src = S > 1 ? static_cast<const void *>(p->idbuf[k]) : static_cast<const void *>(c->gen + (J - 2 * (uint64_t)S));
                if (hipMemcpyAsync(&c->ctl->token, src, sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "id copy failed"); break; }
            }
            if (rank > 0 && hipMemcpyAsync(c->x_in, p->xin[k], c->D * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { rc = fail(RWKV_E_DEVICE, "hop copy failed"); break; }
            rc = run_token(c, lastr);
            if (!rc && rank < S - 1 && hipMemcpyAsync(p->xout[k], c->x, c->D * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = fail(RWKV_E_DEVICE, "hop copy failed");
            if (!rc && hipEventRecord(p->ev_cp[k], c->stream) != hipSuccess) rc = fail(RWKV_E_DEVICE, "event record failed");
        }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    for (int k = 0; k < 2; k++) { const hipError_t e2 = hipStreamSynchronize(p->cs[k]); if (e == hipSuccess) e = e2; }
    if (!rc && e != hipSuccess) rc = fail(RWKV_E_DEVICE, "pipeline decode (dual): %s", hipGetErrorString(e));
    if (!rc && lastr) {
        std::vector<uint64_t> g(2 * n_sub);
        if (hipMemcpy(g.data(), c->gen, 2 * n_sub * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RWKV_E_DEVICE, "copy of the picks failed");
        else
            for (uint64_t i = 0; i < n_sub; i++)
                for (int k = 0; k < 2; k++) picks[(2 * (i % S) + k) * n_steps + i / S] = g[2 * i + k];
    }
    { const int dc = device_check(c); if (!rc) rc = dc; }
    if (!rc && rc_begin) rc = rc_begin;
    if (!rc && bad_id) rc = fail(RWKV_E_ARG, "token id out of range (the schedule ran with id 0 in its place)");
    return rc;
}

int rwkv_pipe_decode(rwkv_ctx *c, const uint64_t *first_tokens, uint64_t n_steps, uint64_t *picks)
{
    if (!c || !c->pipe) return fail(RWKV_E_STATE, "needs a loaded context with rwkv_pipe_init done");
    return rwkv_pipe_decode_streams(c, first_tokens, n_steps, (uint64_t)c->pipe->world, picks);
}
// hop timing (off by default: two event records per tick): out4 = {ticks measured, mean, min, max us} of the span, on this rank's
// stream, from just before the tick's RCCL group {send | recv} to just behind it, over the ticks of the last rwkv_pipe_decode* call
// whose group held a receive.  The MIN is the hop itself (the peer's data was waiting); the mean includes waiting for the peer.
int rwkv_pipe_profile(rwkv_ctx *c, int on)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    c->pipe_prof = on != 0;
    return 0;
}
int rwkv_pipe_hop_stats(rwkv_ctx *c, double *out4)
{
    if (!c || !out4) return fail(RWKV_E_ARG, "NULL argument");
    for (int k = 0; k < 4; k++) out4[k] = c->hop_stats[k];
    return 0;
}

// ---- prompt chunks on a pipeline stage -----------------------------------------------------------------------
// One chunk (n <= 32 tokens of ONE sequence, GPT mode) through this context's layers on the MFMA path; the chunk's
// residual stream sits in buffer `buf` (0/1): stage 0 fills it from `tokens`, a later stage expects the previous stage's
// output there (rwkv_xseq_device / rwkv_xseq_copy, or the RCCL recv of rwkv_pipe_prefill).  Asynchronous.
int rwkv_stage_chunk(rwkv_ctx *c, const uint64_t *tokens, uint64_t n, uint64_t row0, int buf)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded) return fail(RWKV_E_STATE, "RWKV not loaded");
    if (!c->seq_ok) return fail(RWKV_E_STATE, "chunked path not available (load with max_ctx > 1)");
    if (n == 0 || n > (uint64_t)SEQ_TM || row0 + n > c->maxT || (buf != 0 && buf != 1)) return fail(RWKV_E_ARG, "bad chunk (n %llu <= 64, row0 %llu, buf %d)", (unsigned long long)n, (unsigned long long)row0, buf);
    if (c->l0 == 0) {
        if (!tokens) return fail(RWKV_E_ARG, "stage 0 needs the token ids");
        for (uint64_t t = 0; t < n; t++) if (tokens[t] >= RWKV_VOCAB) return fail(RWKV_E_ARG, "token id out of range");
    }
    HIPCHK(hipSetDevice(c->device));
    return enqueue_chunk(c, tokens, (int)n, row0, false, buf);
}
double *rwkv_xseq_device(rwkv_ctx *c, int buf) { return (c && (buf == 0 || buf == 1)) ? c->sq_x[buf] : nullptr; }
int rwkv_sync(rwkv_ctx *c)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return device_check(c);
}
// hand a chunk's residual stream from one stage context to the next ON THE SAME DEVICE (virtual stages: tests, and
// several stages per GPU); ordered behind src's work, and dst's later work is ordered behind the copy
int rwkv_xseq_copy(rwkv_ctx *dst, int dbuf, rwkv_ctx *src, int sbuf, uint64_t rows)
{
    if (!dst || !src || !dst->seq_ok || !src->seq_ok || dst->D != src->D || rows > (uint64_t)SEQ_TM) return fail(RWKV_E_ARG, "bad copy");
    HIPCHK(hipSetDevice(src->device));
    // Stream-ordered, no host wait: dst's stream waits for what src's stream has enqueued so far (the chunk that filled the
    // buffer), copies, and src's stream waits for the copy before it may overwrite the buffer -- so two stages on one GPU run
    // CONCURRENTLY, stage s on chunk c while stage s-1 is on chunk c+1 (each kernel's start-up and tail under the other's stream).
    if (!dst->xs_ev[0]) {
        HIPCHK(hipEventCreateWithFlags(&dst->xs_ev[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&dst->xs_ev[1], hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(dst->xs_ev[0], src->stream));
    HIPCHK(hipStreamWaitEvent(dst->stream, dst->xs_ev[0], 0));
    HIPCHK(hipMemcpyAsync(dst->sq_x[dbuf & 1], src->sq_x[sbuf & 1], rows * src->D * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipEventRecord(dst->xs_ev[1], dst->stream));
    HIPCHK(hipStreamWaitEvent(src->stream, dst->xs_ev[1], 0));
    return 0;
}

// Pipelined prompt ingestion (RWKV::loadContext, rwkv.h:395-413, across the stages): the prompt's 32-token chunks are
// micro-batches, rank r works on chunk t - r at tick t.  Per tick: ONE RCCL group { send the finished chunk's residual
// stream [rows][D] to r+1 | recv the next chunk's from r-1 into the other buffer }, then the chunk through this stage's
// layers.  tokens: the whole prompt (read on rank 0; other ranks only need n_tokens).  Every chunk writes its logits to rows
// [0, rows) of the last stage's logits buffer (position % 32): after the call the buffer holds the LAST chunk's rows, which is
// what RWKV::loadContext's caller reads; returns after the stage's stream has drained.
int rwkv_pipe_prefill(rwkv_ctx *c, const uint64_t *tokens, uint64_t n_tokens)
{
    if (!c) return fail(RWKV_E_ARG, "NULL ctx");
    if (!c->loaded || !c->pipe) return fail(RWKV_E_STATE, "needs a loaded context with rwkv_pipe_init done");
    if (!c->seq_ok) return fail(RWKV_E_STATE, "chunked path not available (load with max_ctx >= 32)");
    if (c->maxT < (uint64_t)SEQ_T) return fail(RWKV_E_ARG, "needs max_ctx >= %d", SEQ_T);
    Pipe *p = c->pipe;
    const int S = p->world, rank = p->rank;
    if (n_tokens == 0 || (rank == 0 && !tokens)) return fail(RWKV_E_ARG, "empty prompt");
    HIPCHK(hipSetDevice(c->device));
    if (c->herr) *c->herr = 0u;
    // k_seq_embed indexes the table with the id: validate the whole prompt on rank 0.  A bad id must not strand the other ranks in
    // their ncclRecv, so the schedule runs with id 0 in its place and the error is reported once it has drained.
    std::vector<uint64_t> clean;
    bool bad_id = false;
    if (rank == 0) {
        for (uint64_t t = 0; t < n_tokens; t++) bad_id = bad_id || tokens[t] >= RWKV_VOCAB;
        if (bad_id) {
            clean.assign(tokens, tokens + n_tokens);
            for (auto &t : clean) if (t >= RWKV_VOCAB) t = 0;
            tokens = clean.data();
        }
    }
    // micro-batch = one weight pass of the stage: 64 rows (two halves per weight fragment) where the contexts allow, else 32
    const uint64_t CH = p->ch;       // agreed by all ranks in rwkv_pipe_init (a per-rank value here could differ between ranks: hang or corruption)
    if (CH == 0) return fail(RWKV_E_STATE, "chunked path not available on every rank (load with max_ctx >= 32)");
    const uint64_t n_chunks = (n_tokens + CH - 1) / CH;
    auto rows_of = [&](uint64_t ci) { return ci + 1 < n_chunks ? CH : n_tokens - ci * CH; };
    auto has_work = [&](int r, uint64_t t) { return t >= (uint64_t)r && t - r < n_chunks; };
    int rc = 0;
    for (uint64_t tick = 0; tick < n_chunks + S - 1 && !rc; tick++) {
        const bool work = has_work(rank, tick);
        const uint64_t ci = tick - rank;
        if (S > 1) {
            int r = p->GroupStart();
            if (!r && rank < S - 1 && has_work(rank + 1, tick)) r = p->Send(c->sq_x[(ci - 1) & 1], rows_of(ci - 1) * c->D, kNcclFloat64, rank + 1, p->comm, c->stream);
            if (!r && rank > 0 && work) r = p->Recv(c->sq_x[ci & 1], rows_of(ci) * c->D, kNcclFloat64, rank - 1, p->comm, c->stream);
            const int r2 = p->GroupEnd();
            if (r || r2) { rc = pipe_fail(p, r ? r : r2, "RCCL hop"); break; }
        }
        if (work) rc = enqueue_chunk(c, rank == 0 ? tokens + ci * CH : nullptr, (int)rows_of(ci), 0, false, (int)(ci & 1));
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e != hipSuccess) rc = fail(RWKV_E_DEVICE, "pipeline prefill: %s", hipGetErrorString(e));
    { const int dc = device_check(c); if (!rc) rc = dc; }
    if (!rc && bad_id) rc = fail(RWKV_E_ARG, "token id out of range (the schedule ran with id 0 in its place)");
    return rc;
}

void rwkv_pipe_free(rwkv_ctx *c)
{
    if (!c || !c->pipe) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    pipe_dual_teardown(c->pipe);
    if (c->pipe->comm2 && c->pipe->CommDestroy) (void)c->pipe->CommDestroy(c->pipe->comm2);
    if (c->pipe->comm && c->pipe->CommDestroy) (void)c->pipe->CommDestroy(c->pipe->comm);
    delete c->pipe;
    c->pipe = nullptr;
    if (c->pipe_ring) { (void)hipHostFree(c->pipe_ring); c->pipe_ring = nullptr; c->pipe_ring_cap = 0; }
    // a later rank read the hop from x_in: any other transport (rwkv_stage_forward callers write rwkv_x_device()) needs the
    // graphs back on c->x, or it would compute on a stale x_in
    if (c->x_in) { c->x_in = nullptr; (void)rebuild_graphs(c); }
}

// Device pointer behind tensor slot `slot` of the reference's `tensors[]` table (rwkv.h:248, enums/enum.h:7-55) where the
// engine keeps that tensor in FILE layout: the f32/f64 vectors, the five state arrays, x and the logits buffer (BUFFER2).
// The uint8 matrices are re-tiled at load (no file-layout copy stays on the device) and the pure scratch slots have no
// counterpart: NULL.  EMBED is a DEVICE pointer here (the reference keeps the table on the host, rwkv.cu:683-684).
void *rwkv_tensor_device(rwkv_ctx *c, int slot)
{
    if (!c || !c->loaded) return nullptr;
    switch (slot) {
    case X: return c->x;
    case EMBED: return c->embed;
    case LAYERNORMS: return c->ln;
    case STATEXY: return c->state[0];
    case STATEAA: return c->state[1];
    case STATEBB: return c->state[2];
    case STATEPP: return c->state[3];
    case STATEDD: return c->state[4];
    case BUFFER2: return c->logits;
    case MIXK: return c->mixk;
    case MIXV: return c->mixv;
    case MIXR: return c->mixr;
    case KR: return c->kr;
    case VR: return c->vr;
    case RR: return c->rr;
    case O1: return c->o1;
    case O2: return c->o2;
    case O3: return c->o3;
    case ATTOUTR: return c->attr;
    case ATTOUTO: return c->atto;
    case FFNMIXK: return c->fmixk;
    case FFNMIXV: return c->fmixr;
    case FFNKR: return c->fkr;
    case FFNVR: return c->fvr;
    case FFNRR: return c->frr;
    case FFNKO: return c->fko;
    case FFNVO: return c->fvo;
    case FFNRO: return c->fro;
    case HEADR: return c->headr;
    case HEADO: return c->heado;
    default: return nullptr;
    }
}

} // extern "C"
