// mallbench.hip -- does the 256 MiB Infinity Cache work as a prefetch buffer for weight streams?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, bool NT>
__global__ void k_read(const u32x4 *__restrict__ p, size_t n16, unsigned *sink)
{
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

static float timed(hipStream_t s, const u32x4 *p, size_t bytes, unsigned *sink, bool nt)
{
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipEventRecord(a, s));
    if (nt) k_read<8, true><<<256, 512, 0, s>>>(p, bytes / 16, sink); else k_read<8, false><<<256, 512, 0, s>>>(p, bytes / 16, sink);
    CHK(hipEventRecord(b, s)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f;
}

int main()
{
    const size_t MB = 1 << 20, total = 4096 * MB;
    u32x4 *p; unsigned *sink;
    CHK(hipMalloc(&p, total)); CHK(hipMalloc(&sink, 4)); CHK(hipMemset(p, 1, total));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto region = [&](int i) { return p + (size_t)i * 128 * MB / 16; };   // disjoint 128 MiB slots
    int slot = 0;
    for (size_t mb : {17, 50, 84}) {
        for (int nt = 0; nt < 2; nt++) {
            float cold = 0, warm = 0, warm2 = 0;
            for (int rep = 0; rep < 4; rep++) {
                const u32x4 *r = region(slot++ % 30);
                cold += timed(s1, r, mb * MB, sink, nt);
                warm += timed(s1, r, mb * MB, sink, nt);
                warm2 += timed(s1, r, mb * MB, sink, nt);
            }
            printf("%3zu MiB %s: cold %.2f us, re-read %.2f us, re-read2 %.2f us\n", mb, nt ? "nt   " : "plain", cold / 4, warm / 4, warm2 / 4);
        }
    }
    // survive an intervening stream of other data?
    for (size_t other : {64, 128, 192, 256}) {
        float t = 0;
        for (int rep = 0; rep < 4; rep++) {
            const u32x4 *r = region(slot++ % 30), *o = region(slot++ % 30);
            timed(s1, r, 84 * MB, sink, false);
            timed(s1, o, other * MB / 2, sink, true); timed(s1, o + other * MB / 32, other * MB / 2, sink, true);
            t += timed(s1, r, 84 * MB, sink, true);
        }
        printf("84 MiB plain-read, then %3zu MiB of other nt traffic, then nt re-read: %.2f us\n", other, t / 4);
    }
    // concurrent prefetch on a second stream while the main stream is HBM-bound on something else
    for (int pfwaves : {1, 2, 4}) {
        float tmain = 0, tnext = 0;
        for (int rep = 0; rep < 4; rep++) {
            const u32x4 *cur = region(slot++ % 30), *nxt = region(slot++ % 30);
            hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, s1));
            k_read<4, false><<<256, 64 * pfwaves, 0, s2>>>(nxt, 67 * MB / 16, sink);        // prefetcher: small
            k_read<8, true><<<256, 512, 0, s1>>>(cur, 84 * MB / 16, sink);                 // main, HBM-bound
            CHK(hipEventRecord(b, s1)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b)); tmain += ms * 1e3f;
            CHK(hipDeviceSynchronize());
            tnext += timed(s1, nxt, 67 * MB, sink, true);
        }
        printf("prefetch 67 MiB with %d wave(s)/CU beside an 84 MiB main read: main %.2f us, then next (prefetched) %.2f us\n",
               pfwaves, tmain / 4, tnext / 4);
    }
    return 0;
}
