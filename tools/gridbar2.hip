// gridbar2.hip -- where does the time of an in-kernel grid barrier go?  Per-phase wall-clock stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k(unsigned *flags, int iters, unsigned long long *acc, int use_fence, int sleep)
{
    unsigned epoch = 0;
    const unsigned nb = gridDim.x;
    unsigned long long t_rel = 0, t_store = 0, t_poll = 0, t_acq = 0, npoll = 0, t_first = 0;
    for (int it = 0; it < iters; it++) {
        __syncthreads();
        epoch++;
        if (threadIdx.x < 64) {
            unsigned long long a = wall_clock64();
            if (use_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned long long b = wall_clock64();
            if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long c = wall_clock64();
            unsigned spins = 0;
            unsigned long long c1 = 0;
            for (;;) {
                bool ok = true;
                for (unsigned i = threadIdx.x; i < nb; i += 64)
                    ok = ok && (int)(__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
                if (spins == 0) c1 = wall_clock64();
                spins++;
                if (__all(ok) || spins > 1000000u) break;
                if (sleep) __builtin_amdgcn_s_sleep(1);
            }
            unsigned long long d = wall_clock64();
            if (use_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            unsigned long long e = wall_clock64();
            t_rel += b - a; t_store += c - b; t_first += c1 - c; t_poll += d - c; t_acq += e - d; npoll += spins;
        }
        __syncthreads();
    }
    if (blockIdx.x == 3 && threadIdx.x == 0) { acc[0] = t_rel; acc[1] = t_store; acc[2] = t_first; acc[3] = t_poll; acc[4] = t_acq; acc[5] = npoll; }
}

// plain-load latency probes: pointer chase through a small buffer with / without agent scope
__global__ void k_lat(unsigned *buf, int n, unsigned long long *out, int scoped)
{
    unsigned idx = 0;
    unsigned long long a = wall_clock64();
    for (int i = 0; i < n; i++)
        idx = scoped ? __hip_atomic_load(&buf[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(&buf[idx]);
    unsigned long long b = wall_clock64();
    out[0] = b - a; out[1] = idx;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int nb = p.multiProcessorCount;
    unsigned *flags; unsigned long long *acc, h[6];
    CHK(hipMalloc(&flags, 4096)); CHK(hipMalloc(&acc, 64));
    for (int fence = 0; fence < 2; fence++)
        for (int sleep = 0; sleep < 2; sleep++) {
            CHK(hipMemset(flags, 0, 4096));
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            CHK(hipEventRecord(e0));
            k<<<nb, 512>>>(flags, iters, acc, fence, sleep);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h, acc, 48, hipMemcpyDeviceToHost));
            printf("fence=%d sleep=%d: %.2f us/barrier | release %.2f store %.2f first-poll %.2f poll-total %.2f acquire %.2f us, polls/barrier %.1f\n", fence, sleep,
                   ms * 1000.0 / iters, h[0] / 100.0 / iters, h[1] / 100.0 / iters, h[2] / 100.0 / iters, h[3] / 100.0 / iters, h[4] / 100.0 / iters, (double)h[5] / iters);
        }
    // latency probes
    unsigned *buf; const int N = 1 << 20;
    CHK(hipMalloc(&buf, N * 4));
    unsigned *hb = (unsigned *)malloc(N * 4);
    for (int i = 0; i < N; i++) hb[i] = (unsigned)(((unsigned long long)i * 40503u + 12345u) % N);
    CHK(hipMemcpy(buf, hb, N * 4, hipMemcpyHostToDevice));
    for (int scoped = 0; scoped < 2; scoped++) {
        k_lat<<<1, 1>>>(buf, 2000, acc, scoped);
        CHK(hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost));
        printf("dependent-load latency (%s): %.0f ns\n", scoped ? "agent-scope" : "plain", h[0] * 10.0 / 2000);
    }
    return 0;
}
