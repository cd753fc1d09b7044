/*
 * rwkv_mi355x.h -- C-ABI of the MI355X-native RWKV-v4 uint8 inference engine
 * (librwkv_mi355x.so).  Plain pointers and sizes only; no C++ or torch types.
 *
 * This is the drop-in boundary that sits where the reference's backend link
 * boundary sits: the seven C++-linkage free functions that reference
 * include/rwkv/rwkv/rwkv.h:63-122 declares and include/rwkv/cuda/rwkv.cu
 * defines.  The 46-pointer interface of the reference is replaced by an opaque
 * handle; each entry point cites the reference function it replaces.
 * The C++ drop-in header include/rwkv.h (class RWKV / RWKVState) and the pybind
 * module `rwkv` are thin wrappers over these calls.
 *
 * All functions return 0 on success and a negative rwkv_status on failure;
 * rwkv_last_error() gives a human-readable message for the calling thread.
 * Nothing here falls back to a CPU path: without a gfx950 device every
 * compute entry point fails with RWKV_E_DEVICE.
 */
#ifndef RWKV_MI355X_H
#define RWKV_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this C-ABI: bumped whenever an existing entry point changes its argument list or meaning (additions alone do
 * not bump it).  3: rwkv_decode_typical / rwkv_sample_typical take `flags`; the one-launch-token hooks (rwkv_one_launch,
 * rwkv_debug_mega_timeline) are gone; bounded device-side waits report RWKV_E_DEVICE.  4: weight rows carried in LDS that fail
 * their check are re-loaded instead of failing the call (no more "arrived damaged" error); the carry counters are on by default
 * (rwkv_debug_carry_stats).  5: the two-communicator pipeline schedule.  6: the weight rows carried in LDS across kernel boundaries
 * are gone and with them rwkv_debug_carry_stats / rwkv_debug_carry_hits; the stand-alone GEMV entry point rwkv_mm8_one -- a kernel that only the tests launched -- is
 * replaced by rwkv_debug_launch / rwkv_debug_read / rwkv_debug_write, which run and inspect the PRODUCTION decode kernels one at a time.
 * rwkv_abi_version() returns the value the
 * library was built with: a binding compares it with the header it was compiled against. */
#define RWKV_MI355X_ABI_VERSION 6
int rwkv_abi_version(void);

#define RWKV_VOCAB 50277u /* hard-wired in the reference: rwkv.h:126, rwkv.cu:471,589 */
#define RWKV_N_TENSORS 46 /* tensor slots of model.bin: enums/enum.h:7-55 */

enum rwkv_mode { RWKV_MODE_PARRALEL = 0, RWKV_MODE_GPT = 1 }; /* enums/enum.h:2-5 (sic) */

enum rwkv_status {
    RWKV_OK = 0,
    RWKV_E_ARG = -1,    /* bad argument (reference: std::runtime_error in rwkv.h:285,344,349) */
    RWKV_E_IO = -2,     /* file missing / truncated (reference: exit(1), rwkv.cu:641-645) */
    RWKV_E_DEVICE = -3, /* HIP error or no usable device (reference: unchecked) */
    RWKV_E_STATE = -4   /* called in the wrong state (not loaded / already loaded) */
};

typedef struct rwkv_ctx rwkv_ctx;

/* State arrays addressable through rwkv_state_device(). */
enum rwkv_state_id { RWKV_STATE_XY = 0, RWKV_STATE_AA = 1, RWKV_STATE_BB = 2, RWKV_STATE_PP = 3, RWKV_STATE_DD = 4 };

/* Create an empty context bound to HIP device `device`.  (No reference counterpart:
 * the reference uses the process-default device.) */
int rwkv_create(rwkv_ctx **out, int device);

/* Replaces load(), rwkv.cu:638-717: read model.bin (2 x u64 header,
 * 46 raw tensors), upload, re-tile the uint8 matrices into the engine's row-per-output
 * layout, allocate scratch and state for `max_ctx` tokens/slots (the reference's maxGPT). */
int rwkv_load_file(rwkv_ctx *ctx, const char *path, uint64_t max_ctx);

/* Same as rwkv_load_file but from memory: `ptrs[i]` is tensor slot i in FILE layout
 * (host memory if on_device == 0, device memory otherwise).  The 13 scratch/state slots
 * may be NULL.  Used by the converter path and by bench.py's synthetic 7B/14B models. */
int rwkv_load_tensors(rwkv_ctx *ctx, uint64_t n_layers, uint64_t n_embed,
                      const void *const *ptrs, int on_device, uint64_t max_ctx);

uint64_t rwkv_n_layers(const rwkv_ctx *ctx);
uint64_t rwkv_n_embed(const rwkv_ctx *ctx);
uint64_t rwkv_max_ctx(const rwkv_ctx *ctx);

/* Replaces cuda_rwkv_parralel(), rwkv.cu:493-593 (and cuda_rwkv, :595-628): run `n_tokens`
 * tokens.  GPT mode: consecutive tokens of one sequence on state slot 0.  PARRALEL mode: one
 * step of n_tokens independent sequences on state slots 0..n_tokens-1.  Logits for every
 * position land in the device logits buffer ([n_tokens][50277] f32); state stays on the device.
 * n_tokens == 1 runs the decode kernels (uint8 x fixed-point dot products on the VALU); n_tokens >= 2
 * on a whole-model context with max_ctx > 1 runs mm8_seq on the int8 matrix cores in weight passes of up
 * to 64 rows (two 32-row halves that share every weight fragment; a call of <= 32 rows, or env
 * RWKV_SEQ_ROWS=32: one 32-row chunk per pass), every weight byte read once per pass, the passes of a
 * longer call pipelined over RWKV_SEQ_STAGES (default 3) streams of the one GPU (env RWKV_SEQ=0 at load
 * time keeps the token-by-token path).  All schedules give bit-identical results and agree with the
 * reference within its own tolerance; see DESIGN.md 5.
 * Synchronous on return (the reference ends with cudaDeviceSynchronize, rwkv.cu:590). */
int rwkv_forward(rwkv_ctx *ctx, const uint64_t *tokens, uint64_t n_tokens, int mode);

/* Replaces setState(), rwkv.cu:479-490: upload host state arrays (each [n_slots][L][D] f64). */
int rwkv_set_state(rwkv_ctx *ctx, const double *xy, const double *aa, const double *bb,
                   const double *pp, const double *dd, uint64_t n_slots);

/* Replaces getOutput(), rwkv.cu:467-477: download logits ([n_tokens][50277] f32) and the five
 * state arrays (each [n_tokens][L][D] f64).  Any pointer may be NULL to skip that copy. */
int rwkv_get_output(rwkv_ctx *ctx, float *logits, double *xy, double *aa, double *bb,
                    double *pp, double *dd, uint64_t n_tokens);

/* Zero all state slots on the device (what `new RWKVState` + setState does, rwkv.h:163-170). */
int rwkv_reset_state(rwkv_ctx *ctx);

/* Device-side greedy continuation (storygen's loop, examples/storygen/storygen.cpp:63-69, with
 * argmax in place of typical(): logit 0 is banned as out[0] = -99 does there).  Feeds
 * `first_token`, then n_tokens-1 times the argmax of the previous logits, all on state slot 0
 * without host round trips; writes the n_tokens picked ids to out_tokens (host).  The logits of
 * the LAST step remain in the device logits buffer (row 0). */
int rwkv_decode_greedy(rwkv_ctx *ctx, uint64_t first_token, uint64_t n_tokens, uint64_t *out_tokens);

/* The reference's sampler typical() (include/rwkv/sampler/typical.h:20-58) on the device, from the logits row
 * `row` of the LAST forward, without downloading them.  The draw is the inverse CDF in token order for the
 * caller's uniform u in [0, 1) (include/rwkv_sampler.h typical_u() is the same draw on the host).
 * Default = what typical.h COMPUTES (pinned by tests/test_sampler_ref_cpu.py against 20 000 draws of the
 * reference's own function per case): a draw from softmax(logits)^n with n = uint8(1/temp) -- its typical-set cut
 * at tau assigns into a temporary and has no effect (typical.h:50) and nc::power takes an integer exponent
 * (typical.h:52); n = 0 (temp > 1) is the uniform distribution.  RWKV_SAMPLE_RECIPE instead applies the recipe its
 * header comment documents (entropy, |-log p - H| ordering, smallest prefix with cumulative probability >= tau,
 * p^(1/temp)).  RWKV_SAMPLE_BAN0 first sets logit 0 to -99 as storygen does (examples/storygen/storygen.cpp:66). */
#define RWKV_SAMPLE_BAN0 1
#define RWKV_SAMPLE_RECIPE 2
int rwkv_sample_typical(rwkv_ctx *ctx, uint64_t row, float temp, float tau, double u, int flags, uint64_t *token);

/* Device-side sampled continuation: storygen's loop (examples/storygen/storygen.cpp:63-69: forward, out[0] = -99,
 * typical) with the sampler above in place of the host typical(); u of step k = uniform(splitmix64(seed + k));
 * flags: RWKV_SAMPLE_RECIPE (logit 0 is always banned here).  Same contract as rwkv_decode_greedy otherwise. */
int rwkv_decode_typical(rwkv_ctx *ctx, uint64_t first_token, uint64_t n_tokens, float temp, float tau,
                        uint64_t seed, int flags, uint64_t *out_tokens);

/* ---- layer pipeline (no reference counterpart: the reference is single-device) ----
 * A context may own a contiguous layer range [l0, l1) of the model: call rwkv_set_layer_range()
 * before loading.  The first stage (l0 == 0) also owns the embedding table + ln0, the last stage
 * (l1 == n_layers) ln_out + head.  rwkv_stage_forward() runs this stage's share of ONE token on
 * state slot `slot`: stage 0 starts from `token`; later stages start from the residual vector the
 * caller has placed in rwkv_x_device() (f64[n_embed], e.g. by an RCCL recv) and leave their output
 * there for the next hop.  On the last stage `pick` (may be NULL) receives the greedy id. */
int rwkv_set_layer_range(rwkv_ctx *ctx, uint64_t l0, uint64_t l1);
int rwkv_stage_forward(rwkv_ctx *ctx, uint64_t token, uint32_t slot, uint64_t *pick);
double *rwkv_x_device(rwkv_ctx *ctx);

/* One prompt chunk (n <= 64 consecutive tokens of one sequence, GPT mode: one weight pass, two 32-row halves above 32) through THIS stage's layers on the mm8_seq /
 * MFMA path.  The chunk's residual stream [n][n_embed] f64 lives in buffer `buf` (0 or 1) of the context
 * (rwkv_xseq_device): stage 0 fills it from `tokens`, a later stage expects the previous stage's output there and leaves
 * its own in it; the last stage also writes the logits rows [row0, row0 + n).  Asynchronous (rwkv_sync to wait). */
int rwkv_stage_chunk(rwkv_ctx *ctx, const uint64_t *tokens, uint64_t n, uint64_t row0, int buf);
double *rwkv_xseq_device(rwkv_ctx *ctx, int buf);
/* Same-device hand-over of a chunk's residual stream between two stage contexts (several stages per GPU, tests). */
int rwkv_xseq_copy(rwkv_ctx *dst, int dst_buf, rwkv_ctx *src, int src_buf, uint64_t rows);
int rwkv_sync(rwkv_ctx *ctx);

/* ---- native transport of the pipeline: RCCL send/recv over xGMI, enqueued on the engine's own stream ----
 * (north_star: "layers optionally pipeline across the 8 GPUs of one node via RCCL send/recv over xGMI").  One process
 * per GPU; rank r's context holds layers [l0_r, l1_r) (rwkv_set_layer_range before loading), rank 0 the embedding, the
 * last rank the head.  librccl.so is resolved at run time by the first of these calls, so single-GPU users never load it.
 *   rwkv_pipe_rccl_path  which RCCL these calls bind in this process (path of the shared object holding ncclSend; loads it if
 *                        need be, no GPU involved).  Order: RWKV_RCCL_LIB if set; else the copy the process has ALREADY loaded (a
 *                        PyTorch-ROCm host carries torch/lib/librccl.so, built against the HIP runtime torch brought along --
 *                        the engine's streams live in that runtime too); else the system's librccl.so.1
 *   rwkv_pipe_unique_id  one rank makes the 128-byte communicator id (ncclGetUniqueId) and distributes it out of band
 *   rwkv_pipe_init       every rank joins (ncclCommInitRank); the rank must match the context's layer range.  The ranks then AGREE,
 *                        over the communicator, on the prefill micro-batch (64 or 32 rows: RWKV_SEQ_ROWS and max_ctx are per-rank
 *                        values) and on n_embed / n_layers / world; a mismatch fails every rank with RWKV_E_ARG.  One line describing
 *                        this rank's end of the transport goes to stderr
 *   rwkv_pipe_info       that description as a JSON object: rank, world, layer range, device ordinal, PCI bus id, arch, RCCL version
 *                        code + path of the shared object holding ncclSend, HIP runtime version, agreed prefill rows
 *   rwkv_pipe_decode     greedy decode of `world` independent streams (one per stage in flight, state slot = stream),
 *                        n_steps tokens each; first_tokens[world] is read on rank 0, picks[world][n_steps] written on
 *                        the last rank.  The hop (f64[n_embed] forward, the picked id u64 back to rank 0) and the stage
 *                        graph alternate on one stream; the host does not wait inside the loop
 *   rwkv_pipe_decode_streams   the same with only the first n_streams (1..world) streams' slots of the schedule filled:
 *                        n_streams = 1 is ONE stream through all the stages -- the latency of single-stream decode on
 *                        the pipeline, t_tok + (world - 1) hops + the fed-back id; picks stays [world][n_steps]
 *   rwkv_pipe_profile / rwkv_pipe_hop_stats   hop timing: event pairs around the per-tick RCCL group of the last
 *                        rwkv_pipe_decode* call; out4 = {ticks measured, mean, min, max microseconds} (min = the hop
 *                        itself, the peer's data was waiting; the mean includes waiting for the peer)
 *   rwkv_pipe_prefill    RWKV::loadContext (rwkv.h:395-413) across the stages: the prompt's 64-token passes (32 with max_ctx < 64 or RWKV_SEQ_ROWS=32) are the
 *                        micro-batches, stage s works on chunk t - s at tick t; needs max_ctx >= 32; tokens read on rank 0 */
int rwkv_pipe_rccl_path(char *out, uint64_t cap);
int rwkv_pipe_unique_id(void *out128);
int rwkv_pipe_init(rwkv_ctx *ctx, const void *id128, int rank, int world);
int rwkv_pipe_info(rwkv_ctx *ctx, char *out, uint64_t cap);
int rwkv_pipe_decode(rwkv_ctx *ctx, const uint64_t *first_tokens, uint64_t n_steps, uint64_t *picks);
int rwkv_pipe_decode_streams(rwkv_ctx *ctx, const uint64_t *first_tokens, uint64_t n_steps, uint64_t n_streams, uint64_t *picks);
/* 2 x world streams in flight on TWO communicators (the second one is created at the first call, its id travels over the first): the
 * streams of even and odd index run the tick schedule above on their own communicator and HIP stream, the stage alternates between
 * them on the engine's stream -- one parity's hop is in flight while the stage works on the other's item (a tick costs
 * max(t_stage, t_hop) instead of their sum).  first_tokens[2 * world] (rank 0), picks[2 * world][n_steps] (last rank); needs
 * max_ctx >= 2 * world state slots.  Every rank of the pipeline must call it. */
int rwkv_pipe_decode_dual(rwkv_ctx *ctx, const uint64_t *first_tokens, uint64_t n_steps, uint64_t *picks);
int rwkv_pipe_profile(rwkv_ctx *ctx, int on);
int rwkv_pipe_hop_stats(rwkv_ctx *ctx, double *out4);
int rwkv_pipe_prefill(rwkv_ctx *ctx, const uint64_t *tokens, uint64_t n_tokens);
void rwkv_pipe_free(rwkv_ctx *ctx);

/* Replaces freeTensors(), rwkv.cu:719-730, plus destruction of the handle. */
void rwkv_free(rwkv_ctx *ctx);

const char *rwkv_last_error(void);

/* ---- introspection / measurement hooks (no reference counterpart) ---- */

/* Device pointer behind slot `slot` of the reference's RWKV::tensors[] table (rwkv.h:248; slots: enums/enum.h:7-55)
 * where the engine keeps that tensor in FILE layout: the f32/f64 vectors, the five state arrays, X and BUFFER2 (logits).
 * NULL for the uint8 matrices (re-tiled at load) and for pure scratch.  EMBED is a device pointer here (the reference
 * keeps the table on the host, rwkv.cu:683-684). */
void *rwkv_tensor_device(rwkv_ctx *ctx, int slot);
/* Device pointer of the logits buffer ([max_ctx][50277] f32). */
float *rwkv_logits_device(rwkv_ctx *ctx);
/* Device pointer of a state array ([max_ctx][L][D] f64), rwkv_state_id. */
double *rwkv_state_device(rwkv_ctx *ctx, int which);
/* The HIP stream (hipStream_t) all engine work is enqueued on. */
void *rwkv_stream(rwkv_ctx *ctx);
/* Device bytes this context holds: decode-layout weights + row sums + site tables + embedding + state + scratch and, when
 * max_ctx > 1 on a whole-model context, the SECOND resident copy of the matrices in the MFMA B-operand image of the chunk
 * path (+7.2 GB at 7B, +13.9 GB at 14B). */
uint64_t rwkv_resident_bytes(const rwkv_ctx *ctx);
/* Which of the four per-layer decode kernel classes of a loaded context stream the tile image (csrc/tile.hip.h):
 * bit 0 K/V/R + WKV, bit 1 att_out, bit 2 ffn k/r, bit 3 ffn_v; a clear bit = that class streams its matrices in row form.  A
 * class's matrices are resident in the one layout its kernel streams (15 at 4096 channels on 256 CUs, 4 at 5120, 0 otherwise;
 * RWKV_TILE=<mask> before the load overrides).  -1 without a loaded model. */
int rwkv_decode_form(const rwkv_ctx *ctx);
/* Algorithmic HBM bytes of one token (13*L*D^2 + V*D uint8 weight bytes
 * + 168*L*D + 40*D bytes of vectors/state). */
uint64_t rwkv_bytes_per_token(const rwkv_ctx *ctx);

/* Per-kernel-class device time of the last rwkv_profile_token() call, measured with HIP events
 * on the engine's stream.  Classes: 0 embed+ln0, 1 att K/V/R+wkv, 2 att_out, 3 ffn r+k,
 * 4 ffn_v, 5 head, 6 argmax.  ms[c] = total milliseconds over `reps` tokens, bytes[c] =
 * algorithmic uint8 weight bytes of one launch of that class, launches[c] = launches/token. */
#define RWKV_N_KCLASS 7
int rwkv_profile_token(rwkv_ctx *ctx, uint64_t token, int reps, double *ms, uint64_t *bytes,
                       uint32_t *launches);

/* Like rwkv_profile_token, but ONE event pair brackets a batch of `reps` x (launches per token) back-to-back
 * launches of each class, so the per-launch figure is not inflated by per-bracket event overhead and is
 * comparable with rocprofv3 kernel durations.  ms[c] = total ms of the batch, n[c] = launches in it.
 * Leaves the recurrent state zeroed (the batches run the kernels out of token order). */
int rwkv_profile_batched(rwkv_ctx *ctx, uint64_t token, int reps, double *ms, uint32_t *n);

/* Tuning aid: run one eager token with phase timestamps enabled in the middle layer's ffn r+k
 * kernel; out receives grid*8*8 stamps of the 100 MHz device wall clock ([workgroup][wave][phase]). */
int rwkv_debug_timeline(rwkv_ctx *ctx, uint64_t token, unsigned long long *out, uint64_t cap);

/* ---- per-kernel parity hooks: the production decode kernels one launch at a time ----
 * rwkv_debug_launch runs ONE launch of decode kernel class `cls` -- 0 k_first (embedding + ln0, rwkv.cu:513-524; opens the first ln1
 * site), 1 k_att (ln1, mixatt, mm8_threec, wkv_forward: rwkv.cu:535-545), 2 k_attout (mm8_one att_out + residual, :548-553),
 * 3 k_ffn_rk (ln2, mixffn, mm8_one ffn_r + sigmoid, mm8_one ffn_k + relu^2, :557-573), 4 k_ffnv (mm8_one<float> ffn_v + blockout,
 * :574-577), 5 k_head (ln_out + mm8_one head, :585-589), 6 the greedy pick -- of layer `layer`, eagerly, in the form (row / tile) the
 * context runs that class in, on whatever its buffers hold, and waits for it.  cls 0 also sets the token's control block (token id, state
 * slot); the other classes take the slot from the last cls-0 call.  A token is 0, then 1..4 per layer, then 5: the launches of the
 * captured token graph, one at a time, so that a test can read every hand-over between two kernels and check each kernel against the
 * oracle's piece for it on that kernel's OWN inputs (tests/test_kernels_gpu.py).
 * rwkv_debug_read / rwkv_debug_write copy an intermediate vector of the decode path to / from host memory (sizes in bytes; n_embed = D,
 * grid = rwkv_debug_grid() workgroups per launch). */
enum rwkv_debug_buf {
    RWKV_DBG_X = 0,        /* f64[D]     residual stream */
    RWKV_DBG_YBUF = 1,     /* f32[D]     k_att -> k_attout: gated wkv output (rwkv.cu:250, cast to f32 as :290 does) times att_out's scale r[j] */
    RWKV_DBG_PART_ATT = 2, /* f64[grid]  k_att -> k_attout: per-workgroup partial sums of (gated wkv) * att_out's offset o[j] */
    RWKV_DBG_PMAX_ATT = 3, /* f32[grid]  k_att -> k_attout: per-workgroup max |YBUF| */
    RWKV_DBG_HBUF = 4,     /* f32[4 D]   k_ffn_rk -> k_ffnv: relu(k)^2 (rwkv.cu:189-190) times ffn_v's scale r[j] */
    RWKV_DBG_RGATE = 5,    /* f32[D]     k_ffn_rk -> k_ffnv: sigmoid(r) (rwkv.cu:212) */
    RWKV_DBG_PART_FFN = 6, /* f64[grid]  k_ffn_rk -> k_ffnv: per-workgroup partial sums of relu(k)^2 * ffn_v's offset o[j] */
    RWKV_DBG_PMAX_FFN = 7, /* f32[grid]  k_ffn_rk -> k_ffnv: per-workgroup max |HBUF| */
    RWKV_DBG_LNSTAT = 8    /* f64[3][2]  mean, rstd of the last ln1 / ln2 / ln_out site */
};
int rwkv_debug_launch(rwkv_ctx *ctx, int cls, uint64_t layer, uint64_t token, uint32_t slot);
int rwkv_debug_read(rwkv_ctx *ctx, int which, void *dst, uint64_t cap_bytes);
int rwkv_debug_write(rwkv_ctx *ctx, int which, const void *src, uint64_t bytes);
uint64_t rwkv_debug_grid(const rwkv_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RWKV_MI355X_H */
