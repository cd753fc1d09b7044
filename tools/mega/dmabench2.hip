// dmabench2.hip -- the one-launch token's REAL loader and consumer group loop (csrc/mega.hip.h: MegaLoader, mg_groups) streaming the
// weights of NL synthetic 7B-shaped layers, with the prologue / hand-off replaced by LDS counter bumps: what the ring protocol
// itself sustains, phase after phase, without any chip-wide synchronisation.  Build with the same -DRWKV_MG_* knobs as the engine.
#include "mega.hip.h"
#include <cstdio>
#include <cstdlib>
using namespace rwkvk;
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <int S, int WORK>
__global__ __launch_bounds__(NT) void k_bench(MegaArgs a, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using LY = MegaLds<S>;
    unsigned *xq = reinterpret_cast<unsigned *>(smem + LY::XQ);
    MegaCtl *mc = reinterpret_cast<MegaCtl *>(smem + LY::CTL);
    unsigned char *ring = smem + LY::RING;
    const int D = a.D, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunks = D >> 4, nq = mg_nq(a);
    if (threadIdx.x < sizeof(MegaCtl) / 4) reinterpret_cast<unsigned *>(mc)[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < 4 * S * 768; i += NT) xq[i] = 0x01010101u * (i & 3);
    __syncthreads();
    if (wave == MG_NC) { MegaLoader<S> ld(a, mc, lds_addr(ring), lane); ld.run(); return; }
    const int cw = wave;
    MegaPos mp;
    unsigned long long acc = 0;
    // phase 0 has no weights
    if (lane == 0) __hip_atomic_fetch_add(&mc->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int q = 1; q < nq; q++) {
        int kind, li, g0, g1;
        mg_decode(a, q, kind, li);
        mg_range(kind, D, g0, g1);
        // stand-in for the edge: all consumer waves of THIS workgroup done with phase q - 1, then "gathered" and "staged"
        mg_wait_ge(&mc->done, (unsigned)q * MG_NC, mc, a);
        if (cw < MG_NP && lane == 0) { __hip_atomic_fetch_add(&mc->gathered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&mc->staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        mg_wait_ge(&mc->staged, (unsigned)q * MG_NP, mc, a);
        auto pre = [&](int) { return 0; };
        const int ng = g1 - g0;
        if (kind == MG_ATT) {
            auto epi = [&](int, const unsigned long long (&T)[3], int) { acc += T[0] + T[1] + T[2]; };
            if (WORK) mg_groups<S, 3, PAT_PER_ROW>(a, ng, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            else mg_groups<S, 3, PAT_PER_ROW>(a, ng, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 3u * ng; mp.kbase += ng;
        } else if (kind == MG_ATTOUT) {
            auto epi = [&](int, const unsigned long long (&T)[MG_AR], int) { acc += T[0]; };
            mg_groups<S, MG_AR, PAT_SHARED>(a, ng, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += (unsigned)MG_AR * ng; mp.kbase += ng;
        } else if (kind == MG_FRK) {
            auto epi = [&](int, const unsigned long long (&T)[5], int) { acc += T[0] + T[4]; };
            mg_groups<S, 5, PAT_FFN_RK>(a, ng, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 5u * ng; mp.kbase += ng;
        } else if (kind == MG_FV) {
            auto epi = [&](int, const unsigned long long (&T)[4], int) { acc += T[0] + T[3]; };
            mg_groups<S, 4, PAT_PER_ROW>(a, ng, mp, ring, mc, xq, lane, cw, chunks, nq, q, pre, epi);
            mp.ubase += 4u * ng; mp.kbase += ng;
        }
        if (lane == 0) __hip_atomic_fetch_add(&mc->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc == 0x12345ull) sink[threadIdx.x] = (unsigned)acc;
}

int main(int argc, char **argv)
{
    const int NL = argc > 1 ? atoi(argv[1]) : 8, D = 4096;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const size_t per_layer = 13ull * D * D;
    uint8_t *w; unsigned *sink; MegaSync *sync; unsigned *herr;
    CHK(hipMalloc(&w, per_layer * NL)); CHK(hipMemset(w, 3, per_layer * NL));
    CHK(hipMalloc(&sink, 4096)); CHK(hipMalloc(&sync, sizeof(MegaSync))); CHK(hipMemset(sync, 0, sizeof(MegaSync)));
    CHK(hipMalloc(&herr, 64)); CHK(hipMemset(herr, 0, 64));
    AttArgs *ha = new AttArgs[NL](); AttOutArgs *ho = new AttOutArgs[NL](); FfnRKArgs *hk = new FfnRKArgs[NL](); FfnVArgs *hv = new FfnVArgs[NL]();
    for (int l = 0; l < NL; l++) {
        uint8_t *b = w + per_layer * l;
        ha[l].w = b; ho[l].w = b + 3ull * D * D; hk[l].w = b + 4ull * D * D; hv[l].w = b + 9ull * D * D;
    }
    MegaArgs a{};
    AttArgs *da; AttOutArgs *dor; FfnRKArgs *dk; FfnVArgs *dv;
    CHK(hipMalloc(&da, sizeof(AttArgs) * NL)); CHK(hipMalloc(&dor, sizeof(AttOutArgs) * NL)); CHK(hipMalloc(&dk, sizeof(FfnRKArgs) * NL)); CHK(hipMalloc(&dv, sizeof(FfnVArgs) * NL));
    CHK(hipMemcpy(da, ha, sizeof(AttArgs) * NL, hipMemcpyHostToDevice)); CHK(hipMemcpy(dor, ho, sizeof(AttOutArgs) * NL, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dk, hk, sizeof(FfnRKArgs) * NL, hipMemcpyHostToDevice)); CHK(hipMemcpy(dv, hv, sizeof(FfnVArgs) * NL, hipMemcpyHostToDevice));
    a.att = da; a.attout = dor; a.frk = dk; a.fv = dv; a.sync = sync; a.herr = herr; a.nl = NL; a.has_head = 0; a.with_argmax = 0; a.D = D; a.tl = nullptr;
    const size_t smem = MegaLds<4>::BYTES;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bench<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipEventRecord(e0));
        k_bench<4, 1><<<grid, NT, smem>>>(a, sink);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned err = 0; CHK(hipMemcpy(&err, herr, 4, hipMemcpyDeviceToHost));
    printf("%d layers: %.1f us = %.2f us per layer, %.2f TB/s, %.1f KB/us per CU, gave up: %u   [THINQ %d PAUSE %d DEPTH %d NODOT %d]\n", NL, best * 1e3, best * 1e3 / NL,
           (double)per_layer * NL / best / 1e9, (double)per_layer * NL / grid / (best * 1e3) / 1024.0, err, RWKV_MG_THINQ, RWKV_MG_PAUSE, RWKV_MG_DEPTH, RWKV_MG_NODOT);
    return 0;
}
