// chainbench.hip -- can consecutive, DEPENDENT decode kernels overlap on gfx950, and how must they be launched?
//
// A chain of K kernels, 256 workgroups x 512 threads each.  Workgroup w of kernel i:
//   1. streams `pre_kb` KB of its weights (nothing it reads there depends on kernel i - 1: the weight prefetch),
//   2. waits on a device counter until ALL workgroups of kernel i - 1 have arrived (one polling lane, bounded),
//   3. streams `post_kb` KB more, spins for a pseudo-random 0..`spread` us (the end-of-kernel spread of the real kernels),
//   4. adds 1 to kernel i's arrival counter.
// Every workgroup logs wall-clock stamps {start, flag seen, end} and the CU it ran on.  Launch modes:
//   0  one stream, ordinary launches (every kernel waits for the previous one to drain: today's token)
//   1  one stream, hipExtAnyOrderLaunch (AQL packets without the barrier bit): do the next kernel's workgroups start on
//      CUs the previous kernel has left, and are they placed strictly behind ALL of its workgroups (deadlock-free with
//      workgroups that own a whole CU)?
//   2  two streams in ping-pong (kernel i on stream i & 1), ordinary launches
//   3  mode 2 captured into a hipGraph and replayed
// usage: chainbench <mode> <lds_kb> <pre_kb> <post_kb> <spread_us> [K=16] [reps=3]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Args {
    unsigned *done;              // [K + 1] arrival counters (done[0] preset to the grid size)
    unsigned long long *log;     // [K][grid][4]
    const unsigned char *w;      // weights of this kernel: grid * (pre_kb + post_kb) KB
    unsigned *sink;
    int idx, pre_kb, post_kb, spread_us, expect;
    unsigned long long timeout;  // wall-clock ticks (100 MHz)
};

__device__ __forceinline__ unsigned stream_kb(const unsigned char *p, int kb, unsigned acc)
{
    // 8 waves x 1 KiB per wave-instruction, 4 loads in flight per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p) + lane;
    for (int k = wave; k + 24 < kb; k += 32) {
        const u32x4 a = __builtin_nontemporal_load(q + (size_t)k * 64), b = __builtin_nontemporal_load(q + (size_t)(k + 8) * 64);
        const u32x4 c = __builtin_nontemporal_load(q + (size_t)(k + 16) * 64), d = __builtin_nontemporal_load(q + (size_t)(k + 24) * 64);
        acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    return acc;
}

__global__ __launch_bounds__(512) void k_chain(Args a)
{
    extern __shared__ unsigned char smem[];
    __shared__ unsigned long long t[4];
    __shared__ unsigned to;
    const unsigned long long t0 = wall_clock64();
    const size_t per = (size_t)(a.pre_kb + a.post_kb) * 1024;
    const unsigned char *mine = a.w + (size_t)blockIdx.x * per;
    unsigned acc = stream_kb(mine, a.pre_kb, 0u);
    smem[threadIdx.x] = (unsigned char)acc;
    if (threadIdx.x == 0) {
        unsigned timed_out = 0;
        const unsigned long long ts = wall_clock64();
        while ((int)(__hip_atomic_load(&a.done[a.idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)a.expect) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - ts > a.timeout) { timed_out = 1; break; }
        }
        t[1] = wall_clock64();
        to = timed_out;
    }
    __syncthreads();
    acc = stream_kb(mine + (size_t)a.pre_kb * 1024, a.post_kb, acc);
    if (a.spread_us > 0) {
        const unsigned h = (blockIdx.x * 2654435761u + (unsigned)a.idx * 40503u) >> 8;
        const unsigned long long wait = (unsigned long long)(h % (unsigned)(a.spread_us * 100));
        const unsigned long long s0 = wall_clock64();
        while (wall_clock64() - s0 < wait) __builtin_amdgcn_s_sleep(1);
    }
    if (acc == 0x12345u) a.sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&a.done[a.idx + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long *l = a.log + ((size_t)a.idx * gridDim.x + blockIdx.x) * 4;
        l[0] = t0; l[1] = t[1]; l[2] = wall_clock64(); l[3] = ((unsigned long long)to << 32) | ((xcc & 15u) << 16) | ((hw >> 8) & 0xffffu);
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) { printf("usage: chainbench mode lds_kb pre_kb post_kb spread_us [K] [reps]\n"); return 2; }
    const int mode = atoi(argv[1]), lds_kb = atoi(argv[2]), pre_kb = atoi(argv[3]), post_kb = atoi(argv[4]), spread = atoi(argv[5]);
    const int K = argc > 6 ? atoi(argv[6]) : 16, reps = argc > 7 ? atoi(argv[7]) : 3;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    const size_t per = (size_t)(pre_kb + post_kb) * 1024, wb = per * grid;
    const int NW = 4;                                   // distinct weight buffers, rotated: no kernel re-reads what is still in L2 / MALL
    unsigned char *w; CHK(hipMalloc(&w, wb * NW + 4096)); CHK(hipMemset(w, 1, wb * NW));
    unsigned *done, *sink; unsigned long long *log;
    CHK(hipMalloc(&done, (K + 1) * 4)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&log, (size_t)K * grid * 32));
    hipStream_t s[2]; CHK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&fork)); CHK(hipEventCreate(&join));
    std::vector<unsigned> hd(K + 1, 0u); hd[0] = (unsigned)grid;
    std::vector<unsigned long long> hl((size_t)K * grid * 4);
    auto launch_all = [&](bool two) -> int {
        for (int i = 0; i < K; i++) {
            Args a{done, log, w + (size_t)(i % NW) * wb, sink, i, pre_kb, post_kb, spread, grid, 300000ull /* 3 ms */};
            hipStream_t st = two ? s[i & 1] : s[0];
            if (mode == 1) hipExtLaunchKernelGGL(k_chain, dim3(grid), dim3(512), (unsigned)(lds_kb * 1024), st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
            else hipLaunchKernelGGL(k_chain, dim3(grid), dim3(512), (size_t)lds_kb * 1024, st, a);
        }
        CHK(hipGetLastError());
        return 0;
    };
    hipGraphExec_t ge = nullptr;
    if (mode == 3) {
        hipGraph_t g;
        CHK(hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed));
        CHK(hipEventRecord(fork, s[0])); CHK(hipStreamWaitEvent(s[1], fork, 0));
        if (launch_all(true)) return 1;
        CHK(hipEventRecord(join, s[1])); CHK(hipStreamWaitEvent(s[0], join, 0));
        CHK(hipStreamEndCapture(s[0], &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    printf("mode %d lds %d KB pre %d post %d KB/WG spread %d us K %d grid %d  (bytes/kernel %.1f MB)\n", mode, lds_kb, pre_kb, post_kb, spread, K, grid, wb / 1e6);
    for (int r = 0; r < reps; r++) {
        CHK(hipMemcpy(done, hd.data(), (K + 1) * 4, hipMemcpyHostToDevice));
        CHK(hipMemset(log, 0, (size_t)K * grid * 32));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, s[0]));
        if (mode == 3) CHK(hipGraphLaunch(ge, s[0]));
        else {
            if (mode == 2) { CHK(hipEventRecord(fork, s[0])); CHK(hipStreamWaitEvent(s[1], fork, 0)); }
            if (launch_all(mode == 2)) return 1;
            if (mode == 2) { CHK(hipEventRecord(join, s[1])); CHK(hipStreamWaitEvent(s[0], join, 0)); }
        }
        CHK(hipEventRecord(e1, s[0]));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(hl.data(), log, hl.size() * 8, hipMemcpyDeviceToHost));
        int timeouts = 0, early_place = 0;
        unsigned long long base = ~0ull;
        for (size_t i = 0; i < hl.size(); i += 4) if (hl[i]) base = std::min(base, hl[i]);
        double sum_period = 0, sum_ov = 0, sum_wait = 0, sum_startspread = 0;
        for (int i = 0; i < K; i++) {
            unsigned long long smin = ~0ull, smax = 0, emin = ~0ull, emax = 0, fmax = 0;
            for (int b = 0; b < grid; b++) {
                const unsigned long long *l = &hl[((size_t)i * grid + b) * 4];
                smin = std::min(smin, l[0]); smax = std::max(smax, l[0]); emin = std::min(emin, l[2]); emax = std::max(emax, l[2]); fmax = std::max(fmax, l[1]);
                timeouts += (int)(l[3] >> 32);
            }
            if (i > 0) {
                unsigned long long pemax = 0, pemin = ~0ull, psmax = 0;
                for (int b = 0; b < grid; b++) { const unsigned long long *l = &hl[((size_t)(i - 1) * grid + b) * 4]; pemax = std::max(pemax, l[2]); pemin = std::min(pemin, l[2]); psmax = std::max(psmax, l[0]); }
                sum_ov += ((double)pemax - (double)smin) / 100.0;        // > 0: kernel i started before kernel i - 1 had ended
                sum_period += ((double)emax - (double)pemax) / 100.0;
                sum_wait += ((double)fmax - (double)pemax) / 100.0;      // last arrival -> flag seen by the slowest poller
                if (smin < psmax) early_place++;                          // a workgroup of kernel i ran before ALL of kernel i - 1 had been placed
            }
            sum_startspread += ((double)smax - (double)smin) / 100.0;
            if (r == reps - 1 && (i < 3 || i == K - 1))
                printf("  k%-2d start %7.2f..%7.2f  flag<= %7.2f  end %7.2f..%7.2f us\n", i, (smin - base) / 100.0, (smax - base) / 100.0, (fmax - base) / 100.0, (emin - base) / 100.0, (emax - base) / 100.0);
        }
        printf("rep %d: %.1f us total, %.2f us/kernel (event), period %.2f us, overlap(start before prev end) %.2f us, last-arrival->flag %.2f us, start spread %.2f us, timeouts %d, early-placed kernels %d\n",
               r, ms * 1e3, ms * 1e3 / K, sum_period / (K - 1), sum_ov / (K - 1), sum_wait / (K - 1), sum_startspread / K, timeouts, early_place);
    }
    return 0;
}
