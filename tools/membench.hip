// membench.hip -- pure HBM read-bandwidth probe for the GEMV streaming pattern (tuning aid).
// Each wave reads `chunk` contiguous bytes per iteration as 16-B/lane loads with K loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, bool NT>
__global__ void k_read(const u32x4 *__restrict__ p, size_t n16, unsigned *sink)
{
    // wave w reads pieces [w*K*64, (w+1)*K*64) then strides by total waves
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    unsigned acc = 0;
    for (size_t base = wave * K * 64; base + K * 64 <= n16; base += nw * K * 64) {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = NT ? __builtin_nontemporal_load(p + base + k * 64 + lane) : p[base + k * 64 + lane];
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int K, bool NT> double run(const u32x4 *p, size_t bytes, int blocks, int threads, unsigned *sink)
{
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    k_read<K, NT><<<blocks, threads>>>(p, bytes / 16, sink);
    CHK(hipDeviceSynchronize());
    const int reps = 5;
    CHK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) k_read<K, NT><<<blocks, threads>>>(p, bytes / 16, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return bytes * (double)reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 2048) * (size_t)1 << 20;
    u32x4 *p; unsigned *sink;
    CHK(hipMalloc(&p, bytes)); CHK(hipMalloc(&sink, 4));
    CHK(hipMemset(p, 1, bytes));
    printf("buffer %zu MiB\n", bytes >> 20);
    const int cfg[][2] = {{256, 256}, {256, 512}, {256, 1024}, {512, 256}, {512, 512}, {1024, 256}, {2048, 256}, {4096, 256}};
    for (auto &c : cfg) {
        printf("blocks %4d x %4d thr: ", c[0], c[1]);
        printf(" K4 %6.0f", run<4, false>(p, bytes, c[0], c[1], sink));
        printf(" K8 %6.0f", run<8, false>(p, bytes, c[0], c[1], sink));
        printf(" K16 %6.0f", run<16, false>(p, bytes, c[0], c[1], sink));
        printf(" | nt K4 %6.0f", run<4, true>(p, bytes, c[0], c[1], sink));
        printf(" K8 %6.0f", run<8, true>(p, bytes, c[0], c[1], sink));
        printf(" K16 %6.0f GB/s\n", run<16, true>(p, bytes, c[0], c[1], sink));
    }
    // small-buffer (one layer's 84 MB) single-launch timings: what one GEMV-sized read costs
    for (size_t mb : {17, 50, 84, 206}) {
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        float best = 1e9;
        for (int i = 0; i < 20; i++) {
            const u32x4 *q = p + ((size_t)(i % 8) * (256 << 20)) / 16;   // rotate through the buffer to stay out of the caches
            CHK(hipEventRecord(a));
            k_read<8, true><<<256, 512>>>(q, (mb << 20) / 16, sink);
            CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("single launch %3zu MiB: %.2f us -> %.0f GB/s\n", mb, best * 1e3, (mb << 20) / (best * 1e-3) / 1e9);
    }
    return 0;
}
