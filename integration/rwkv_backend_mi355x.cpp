// rwkv_backend_mi355x.cpp -- the reference-side translation unit a maintainer of
// harrisonvanderbyl/rwkv-cpp-accelerated adds (as include/rwkv/mi355x/rwkv.cpp) to link the reference's OWN, UNMODIFIED
// host header include/rwkv/rwkv/rwkv.h and apps against this engine instead of include/rwkv/cuda/rwkv.cu or
// include/rwkv/vulkan/rwkv.cpp: it defines exactly the C++-linkage backend functions that header declares
// (rwkv.h:63-122: load, setState, getOutput, freeTensors, cuda_rwkv, cuda_rwkv_parralel) on top of the C-ABI of
// librwkv_mi355x.so (include/rwkv_mi355x.h).
//
//   g++ -std=c++17 examples/storygen/storygen.cpp <this file> -Iinclude -I<engine>/include -L<engine>/csrc -lrwkv_mi355x
//
// Like rwkv.cu (which only includes enums/enum.h, rwkv.cu:14) it does not include rwkv.h itself: that header defines
// non-inline functions (getSize, Mtypes, getName) that may live in one translation unit only.
// The device tensors stay behind an engine handle PER MODEL: every RWKV object owns its tensors[] table (rwkv.h:248,288; the
// pybind module hands out a fresh RWKV per initRwkv, c_binding.cpp:28-33), so load() creates one engine context per call and
// parks it in the table it is given -- ptrs[X] and ptrs[STATEXY] point to a small tagged block that holds the context; the
// header passes exactly those two slots back into every other backend function (tensors[X] as `x` of cuda_rwkv_parralel,
// tensors[STATEXY] as the first state pointer of setState / getOutput, the whole table to freeTensors).  The remaining slots
// are filled with the device pointers the engine keeps in file layout (rwkv_tensor_device; NULL for the re-tiled matrices):
// RWKV only ever passes them back, and the 44 tensor arguments of the forward call are ignored.
// Compiled and linked by oracle/Makefile (targets storygen_l2, vectordb_l2, terminalchat_l2, two_models_l2) and exercised
// by tests/test_dropin_*.py.
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <string>
#include <tuple>

#include "rwkv/enums/enum.h"   // the reference's MODE + slot enum
#include "rwkv_mi355x.h"

namespace {

struct Parked {                       // what ptrs[X] / ptrs[STATEXY] point to
    uint64_t magic;
    rwkv_ctx *ctx;
};
constexpr uint64_t kMagic = 0x4d49333535585f52ull;   // "R_X553IM"

void die_on(int rc)
{
    if (rc != RWKV_OK) {   // the reference's backend has no error channel either: missing file -> message + exit(1), rwkv.cu:641-645
        std::cout << rwkv_last_error() << std::endl;
        exit(1);
    }
}

rwkv_ctx *ctx_of(const void *slot)
{
    const Parked *p = static_cast<const Parked *>(slot);
    if (!p || p->magic != kMagic || !p->ctx) {
        std::cout << "rwkv_backend_mi355x: this pointer did not come from load() (tensors[X] / tensors[STATEXY] carry the engine handle)" << std::endl;
        exit(1);
    }
    return p->ctx;
}

} // namespace

std::tuple<unsigned long long, unsigned long long> load(const std::string &filename, int **ptrs, unsigned long long maxGPT)
{
    rwkv_ctx *ctx = nullptr;
    const char *dev = getenv("RWKV_DEVICE");
    die_on(rwkv_create(&ctx, dev ? atoi(dev) : 0));
    die_on(rwkv_load_file(ctx, filename.c_str(), maxGPT));
    for (int i = 0; i < 46; i++) ptrs[i] = static_cast<int *>(rwkv_tensor_device(ctx, i));
    Parked *park = new Parked{kMagic, ctx};
    ptrs[X] = reinterpret_cast<int *>(park);
    ptrs[STATEXY] = reinterpret_cast<int *>(park);
    std::cout << "n_layers: " << rwkv_n_layers(ctx) << std::endl << "n_embed: " << rwkv_n_embed(ctx) << std::endl;   // rwkv.cu:653-654
    return std::make_tuple((unsigned long long)rwkv_n_layers(ctx), (unsigned long long)rwkv_n_embed(ctx));
}

// rwkv.cu:479-490.  (The header's callers pass (num_layers, num_embed) into (n_embed, n_layers): only the product matters.)
// dev_xy = tensors[STATEXY] = the parked handle
void setState(unsigned long long, unsigned long long, double *dev_xy, double *, double *, double *, double *,
              double *xy, double *aa, double *bb, double *pp, double *dd, unsigned long long tokenlength)
{
    die_on(rwkv_set_state(ctx_of(dev_xy), xy, aa, bb, pp, dd, tokenlength));
}

// rwkv.cu:467-477; dev_xy = tensors[STATEXY]
void getOutput(unsigned long long, unsigned long long, float *, double *dev_xy, double *, double *, double *, double *,
               float *logitsout, double *xy, double *aa, double *bb, double *pp, double *dd, unsigned long long tokenlength)
{
    die_on(rwkv_get_output(ctx_of(dev_xy), logitsout, xy, aa, bb, pp, dd, tokenlength));
}

// rwkv.cu:719-730
void freeTensors(int **ptrs)
{
    Parked *park = reinterpret_cast<Parked *>(ptrs[X]);
    rwkv_free(ctx_of(park));
    park->magic = 0; park->ctx = nullptr;
    delete park;
    for (int i = 0; i < 46; i++) ptrs[i] = nullptr;
}

// rwkv.cu:493-593; x = tensors[X] = the parked handle
void cuda_rwkv_parralel(unsigned long long, unsigned long long, unsigned long long *token, double *x,
                        float *, double *,
                        double *, double *, double *, double *, double *,
                        double *, float *, float *, float *,
                        double *, double *, double *,
                        uint8_t *, uint8_t *, uint8_t *,
                        float *, float *, float *,
                        float *, float *, float *,
                        uint8_t *, float *, float *,
                        double *, double *,
                        uint8_t *, uint8_t *, uint8_t *,
                        float *, float *, float *,
                        float *, float *, float *,
                        double *, double *, float *,
                        double *, double *,
                        uint8_t *, float *, float *,
                        unsigned long long tokenlength, MODE mode)
{
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "token ids are 64-bit");
    die_on(rwkv_forward(ctx_of(x), reinterpret_cast<const uint64_t *>(token), tokenlength, mode == PARRALEL ? RWKV_MODE_PARRALEL : RWKV_MODE_GPT));
}

// rwkv.cu:595-628
void cuda_rwkv(unsigned long long n_layers, unsigned long long n_emb, unsigned long long token, double *x,
               float *embed, double *layernorms,
               double *statexy, double *stateaa, double *statebb, double *statepp, double *statedd,
               double *buffer1, float *buffer2, float *buffer3, float *buffer4,
               double *mixk, double *mixv, double *mixr,
               uint8_t *km, uint8_t *vm, uint8_t *rm,
               float *kr, float *vr, float *rr,
               float *o1, float *o2, float *o3,
               uint8_t *attout, float *attoutr, float *attouto,
               double *ffnmixk, double *ffnmixv,
               uint8_t *ffnk, uint8_t *ffnv, uint8_t *ffnr,
               float *ffnkr, float *ffnvr, float *ffnrr,
               float *ffnko, float *ffnvo, float *ffnro,
               double *ffnkbuffer, double *ffnvbuffer, float *ffnrbuffer,
               double *decay, double *bonus,
               uint8_t *head, float *headr, float *heado)
{
    cuda_rwkv_parralel(n_layers, n_emb, &token, x, embed, layernorms, statexy, stateaa, statebb, statepp, statedd,
                       buffer1, buffer2, buffer3, buffer4, mixk, mixv, mixr, km, vm, rm, kr, vr, rr, o1, o2, o3,
                       attout, attoutr, attouto, ffnmixk, ffnmixv, ffnk, ffnv, ffnr, ffnkr, ffnvr, ffnrr, ffnko, ffnvo, ffnro,
                       ffnkbuffer, ffnvbuffer, ffnrbuffer, decay, bonus, head, headr, heado, 1, GPT);
}
