// gridbar.hip -- microbenchmark: cost and coherence of a hand-written grid barrier on MI355X
// (one 512-thread workgroup per CU, release/acquire at agent scope, bounded spin).
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o tools/gridbar ; run: tools/gridbar [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

struct Bar { unsigned count; unsigned abort; };

__device__ __forceinline__ bool grid_sync(Bar *b, unsigned &target, unsigned nblk)
{
    __syncthreads();                       // all waves' stores issued and complete (vmcnt(0) for stores)
    bool ok = true;
    if (threadIdx.x == 0) {
        target += nblk;
        __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(&b->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 2000000u || __hip_atomic_load(&b->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&b->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope default for this builtin; agent-scope variant below
    }
    __syncthreads();
    return ok;
}

// variant: fence placement as the engine would use it (explicit agent-scope fences)
__device__ __forceinline__ void grid_sync2(Bar *b, unsigned &target, unsigned nblk)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nblk;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(&b->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000u) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_bar(Bar *b, unsigned base, int iters, double *slots, unsigned *errs, int mode)
{
    unsigned target = base;
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; it++) {
        // every thread writes a value derived from (block, it); after the barrier read another block's values
        slots[(size_t)blockIdx.x * 512 + threadIdx.x] = (double)(it * 1000 + blockIdx.x);
        if (mode == 0) { if (!grid_sync(b, target, nb)) return; }
        else grid_sync2(b, target, nb);
        const unsigned ob = (blockIdx.x + 37 + it) % nb;
        const double v = slots[(size_t)ob * 512 + threadIdx.x];
        if (v != (double)(it * 1000 + ob)) atomicAdd(errs, 1u);
        if (mode == 0) { if (!grid_sync(b, target, nb)) return; }   // WAR: nobody overwrites before all have read
        else grid_sync2(b, target, nb);
    }
}

// variant 3: flag barrier without read-modify-write atomics.  arrive = one agent-scope store of the
// epoch into this block's flag; wait = wave 0 polls all flags (4 per lane) until every flag >= epoch.
template <bool FENCES>
__device__ __forceinline__ void grid_sync3(unsigned *flags, unsigned &epoch, unsigned nblk)
{
    __syncthreads();
    epoch++;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            for (unsigned i = threadIdx.x; i < nblk; i += 64)
                ok = ok && (int)(__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
            if (__all(ok) || ++spins > 1000000u) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <bool FENCES>
__global__ __launch_bounds__(512) void k_bar3(unsigned *flags, unsigned base, int iters, double *slots, unsigned *errs)
{
    unsigned epoch = base;
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; it++) {
        slots[(size_t)blockIdx.x * 512 + threadIdx.x] = (double)(it * 1000 + blockIdx.x);
        grid_sync3<FENCES>(flags, epoch, nb);
        const unsigned ob = (blockIdx.x + 37 + it) % nb;
        const double v = slots[(size_t)ob * 512 + threadIdx.x];
        if (v != (double)(it * 1000 + ob)) atomicAdd(errs, 1u);
        grid_sync3<FENCES>(flags, epoch, nb);
    }
}

// barrier with no payload (pure latency)
__global__ __launch_bounds__(512) void k_bar_only(Bar *b, unsigned base, int iters)
{
    unsigned target = base;
    for (int it = 0; it < iters; it++) grid_sync2(b, target, gridDim.x);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int nb = p.multiProcessorCount;
    Bar *b; double *slots; unsigned *errs;
    CHK(hipMalloc(&b, sizeof(Bar))); CHK(hipMemset(b, 0, sizeof(Bar)));
    CHK(hipMalloc(&slots, (size_t)nb * 512 * 8)); CHK(hipMalloc(&errs, 4)); CHK(hipMemset(errs, 0, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned base = 0;
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            k_bar<<<nb, 512>>>(b, base, iters, slots, errs, mode);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            base += (unsigned)(2 * iters * nb);
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr; Bar hb;
            CHK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
            printf("mode %d rep %d: %d blocks, %d iters x 2 barriers: %.3f ms -> %.2f us per barrier(+store/load); mismatches %u abort %u count %u (expect %u)\n",
                   mode, rep, nb, iters, ms, ms * 1000.0 / (2 * iters), herr, hb.abort, hb.count, base);
            if (hb.abort) return 2;
        }
    }
    unsigned *flags; CHK(hipMalloc(&flags, 4096)); CHK(hipMemset(flags, 0, 4096));
    unsigned ep = 0;
    for (int f = 0; f < 2; f++)
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipMemset(errs, 0, 4));
            CHK(hipEventRecord(e0));
            if (f) k_bar3<true><<<nb, 512>>>(flags, ep, iters, slots, errs); else k_bar3<false><<<nb, 512>>>(flags, ep, iters, slots, errs);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            ep += 2 * iters;
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr; CHK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost));
            printf("flag barrier fences=%d rep %d: %.2f us per barrier(+store/load); mismatches %u\n", f, rep, ms * 1000.0 / (2 * iters), herr);
        }
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        k_bar_only<<<nb, 512>>>(b, base, iters * 2);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        base += (unsigned)(2 * iters * nb);
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("barrier only rep %d: %.2f us per barrier\n", rep, ms * 1000.0 / (2 * iters));
    }
    return 0;
}
