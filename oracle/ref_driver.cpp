// ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" shim over the *reference's own* host class and kernel file,
// compiled from where they lie under /root/reference (never copied into this
// repo) by oracle/Makefile into oracle/_ref/libref.so:
//     reference include/rwkv/cuda/rwkv.cu   (hipcc, via its own #define shim rwkv.cu:1-12)
//   + reference include/rwkv/rwkv/rwkv.h    (class RWKV / RWKVState, rwkv.h:140-429)
// It is the on-device oracle ("the reference's own kernel") that the HIP engine
// and the CPU restatement are pinned against on the MI355X box, and the
// reference-GPU baseline of bench.py's notes.  Needs a GPU to run.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <stdexcept>
#include "rwkv/rwkv/rwkv.h"

extern "C" {

// RWKV::loadFile (rwkv.h:281) -- model.bin on disk
void *ref_load_file(const char *path, uint64_t maxGPT)
{
    RWKV *m = new RWKV();
    m->loadFile(path, maxGPT);
    return m;
}

// Borrow 46 tensors that already live in file layout: device pointers for every
// weight slot, a HOST pointer for EMBED (the reference keeps the table on the
// host, rwkv.cu:683-684).  The 13 scratch/state slots are allocated here
// (x maxGPT, as rwkv.cu:701-706 does).  Handles made this way are never freed
// through ~RWKV (it would hipFree memory it does not own).
void *ref_from_ptrs(uint64_t L, uint64_t D, void **ptrs, uint64_t maxGPT)
{
    RWKV *m = new RWKV();
    const int buffers[13] = {X, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD, BUFFER1, BUFFER2,
                             BUFFER3, BUFFER4, FFNKBUFFER, FFNVBUFFER, FFNRBUFFER};
    for (int i = 0; i < 46; i++) m->tensors[i] = (int *)ptrs[i];
    for (int b = 0; b < 13; b++) {
        void *p = nullptr;
        size_t bytes = getSize(buffers[b], L, D) * types[buffers[b]] * maxGPT;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, bytes);
        m->tensors[buffers[b]] = (int *)p;
    }
    m->num_layers = L;
    m->num_embed = D;
    m->state = new RWKVState(L, D, maxGPT);
    m->statexy = m->state->statexy; m->stateaa = m->state->stateaa; m->statebb = m->state->statebb;
    m->statepp = m->state->statepp; m->statedd = m->state->statedd;
    m->out = new float[50277 * maxGPT]();
    m->maxContext = maxGPT;
    m->ready = true;
    return m;
}

uint64_t ref_n_layers(void *h) { return static_cast<RWKV *>(h)->num_layers; }
uint64_t ref_n_embed(void *h) { return static_cast<RWKV *>(h)->num_embed; }

// RWKV::forward(std::vector<u64>, MODE) (rwkv.h:339-376): state upload, cuda_rwkv_parralel, download.
// Returns RWKV::out ([T][50277] floats, valid until the next call) or NULL on a thrown error.
const float *ref_forward(void *h, const uint64_t *tokens, uint64_t T, int mode)
{
    RWKV *m = static_cast<RWKV *>(h);
    try {
        std::vector<unsigned long long> v(tokens, tokens + T);
        return m->forward(v, mode == 0 ? PARRALEL : GPT);
    } catch (const std::exception &) {
        return nullptr;
    }
}

// host-authoritative state arrays (rwkv.h:143-147): which = 0..4 -> xy, aa, bb, pp, dd
double *ref_state(void *h, int which)
{
    RWKVState *s = static_cast<RWKV *>(h)->state;
    double *p[5] = {s->statexy, s->stateaa, s->statebb, s->statepp, s->statedd};
    return (which >= 0 && which < 5) ? p[which] : nullptr;
}

void ref_free_file_model(void *h) { delete static_cast<RWKV *>(h); }

} // extern "C"
