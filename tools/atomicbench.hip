// atomicbench.hip -- what do fire-and-forget device-scope atomics cost UNDER a running weight stream on gfx950?
//
// The question behind it (round 4): decode keeps four all-to-all launches per layer because att_out and ffn_v need the whole gated-WKV /
// hidden vector.  They do not on their INPUT side: the workgroup that finishes channels [16b, 16b+16) holds the 16 inputs of rows
// [16b, 16b+16) of att.output.weight (file layout w[j][k]), so it can form a split-K partial of ALL D outputs -- if 256 workgroups can
// add 4096 partials each into one accumulator without stalling the stream (the reference does exactly that with float atomicAdd,
// rwkv.cu:292,548-553).  256 workgroups x 512 threads; wave 7 streams `kib` KiB of the workgroup's weights into an LDS ring by DMA
// (issue only, depth 32: dmabench mode 0), waves 0..6 issue the atomics:
//   op 0 none | 1 u64 add | 2 u32 add | 3 f32 add | 4 f64 add | 5 plain 8-byte stores to a private slab (the no-atomics alternative:
//   256 slabs, somebody must add them up) | 6 u64 add, two 32-bit partials packed per word (half the addresses)
//   when 0: all at kernel start | 1: spread over the stream (sleep between instructions) | 2: behind the stream (the fused kernels' tail)
//   rot: workgroup b starts at instruction (b * rot) % n, so that the workgroups do not walk the lines in lock step
// usage: atomicbench [kib=256] [naddr=4096] [launches=24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
constexpr int S = 4, NU = 26, NC = 7;

__device__ __forceinline__ void dma_unit4(const uint8_t *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Args {
    const uint8_t *src; size_t per_wg;
    unsigned long long *acc64; unsigned *acc32; float *accf; double *accd; unsigned long long *slab;
    unsigned long long *tlog;      // [grid][2]: loader done, atomics issued (wall clock, 100 MHz)
    int op, when, rot, naddr, sleep;
};

__global__ __launch_bounds__(512) void k(Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    unsigned char *ring = smem + 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x == 0) *flag = 0u;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (wave == NC) {
        const unsigned nunits = (unsigned)(a.per_wg / (S * 1024));
        const unsigned ringa = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)ring;
        unsigned pos = 0;
        const uint8_t *p = a.src + (size_t)blockIdx.x * a.per_wg + lane * 16;
        for (unsigned u = 0; u < nunits; u++) {
            wait_vm<32 - S>();
            dma_unit4(p + (size_t)u * (S * 1024), (unsigned)__builtin_amdgcn_readfirstlane((int)(ringa + pos * (unsigned)(S * 1024))));
            pos = pos + 1 == (unsigned)NU ? 0u : pos + 1;
        }
        wait_vm<0>();
        if (lane == 0) { __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); a.tlog[blockIdx.x * 2] = wall_clock64() - t0; }
        return;
    }
    if (a.op == 0) return;
    if (a.when == 2) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(2);
    const int per = a.op == 2 || a.op == 3 ? 64 : 64;       // addresses per wave instruction
    const int n = a.naddr / per;                             // instructions of the workgroup
    const int start = (int)(((unsigned)blockIdx.x * (unsigned)a.rot) % (unsigned)n);
    for (int i = wave; i < n; i += NC) {
        int ii = i + start; ii = ii >= n ? ii - n : ii;
        const int addr = ii * per + lane;
        const unsigned v = (unsigned)(addr & 1023) + 1u;
        switch (a.op) {
        case 1: (void)__hip_atomic_fetch_add(a.acc64 + addr, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case 2: (void)__hip_atomic_fetch_add(a.acc32 + addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case 3: (void)__hip_atomic_fetch_add(a.accf + addr, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case 4: (void)__hip_atomic_fetch_add(a.accd + addr, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case 5: a.slab[(size_t)blockIdx.x * a.naddr + addr] = v; break;
        default: (void)__hip_atomic_fetch_add(a.acc64 + addr, ((unsigned long long)v << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        }
        if (a.when == 1) for (int s = 0; s < a.sleep; s++) __builtin_amdgcn_s_sleep(32);
    }
    if (wave == 0 && lane == 0) a.tlog[blockIdx.x * 2 + 1] = wall_clock64() - t0;
}

// the consumer side: every workgroup reads the whole accumulator (the next kernel's prologue), 8 B per thread per pass
__global__ __launch_bounds__(512) void k_read(const unsigned long long *acc, int n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += 512) s += acc[i];
    if (s == 0x1234567ull) out[blockIdx.x] = s;
}

int main(int argc, char **argv)
{
    const size_t kib = argc > 1 ? atoi(argv[1]) : 256;
    const int naddr = argc > 2 ? atoi(argv[2]) : 4096;
    const int launches = argc > 3 ? atoi(argv[3]) : 24;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const size_t per_wg = kib << 10, wb = per_wg * grid;
    const int NB = 6;                                                   // weight buffers in rotation: nothing is re-read from L2 / MALL
    uint8_t *src; CHK(hipMalloc(&src, wb * NB + (1u << 20))); CHK(hipMemset(src, 1, wb * NB));
    Args a{};
    CHK(hipMalloc(&a.acc64, naddr * 8)); CHK(hipMalloc(&a.acc32, naddr * 4)); CHK(hipMalloc(&a.accf, naddr * 4)); CHK(hipMalloc(&a.accd, naddr * 8));
    CHK(hipMalloc(&a.slab, (size_t)grid * naddr * 8)); CHK(hipMalloc(&a.tlog, grid * 16));
    const size_t smem = 1024 + (size_t)NU * S * 1024;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%d workgroups x %zu KiB streamed, %d addresses per workgroup, %d launches per measurement\n", grid, kib, naddr, launches);
    const char *opn[] = {"none", "u64 add", "u32 add", "f32 add", "f64 add", "plain 8-B stores (slab)", "u64 add, 2 x 32 packed"};
    const char *whn[] = {"at start", "spread", "behind the stream"};
    auto run = [&](int op, int when, int rot, int na, int sleep) -> int {
        a.src = src; a.per_wg = per_wg; a.op = op; a.when = when; a.rot = rot; a.naddr = na; a.sleep = sleep;
        CHK(hipMemsetAsync(a.acc64, 0, naddr * 8, st)); CHK(hipMemsetAsync(a.acc32, 0, naddr * 4, st));
        float best = 1e9f;
        int total = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipStreamSynchronize(st));
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; i++) { a.src = src + (size_t)(i % NB) * wb; hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, st, a); }
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            total += launches;
        }
        CHK(hipGetLastError());
        std::vector<unsigned long long> tl(grid * 2);
        CHK(hipMemcpy(tl.data(), a.tlog, grid * 16, hipMemcpyDeviceToHost));
        double ld = 0, at = 0;
        for (int b = 0; b < grid; b++) { ld += tl[b * 2] / 100.0; at += tl[b * 2 + 1] / 100.0; }
        int bad = 0;
        if (op == 1 || op == 6) {
            std::vector<unsigned long long> h(na);
            CHK(hipMemcpy(h.data(), a.acc64, na * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < na; i++) {
                const unsigned long long v = (unsigned long long)((i & 1023) + 1) * grid * total;
                if (h[i] != (op == 1 ? v : ((v << 32) | v))) bad++;
            }
        }
        printf("op %d %-26s %-18s rot %2d n %5d  %7.2f us/launch  (loader done %5.2f us, atomics issued %5.2f us)%s\n", op, opn[op], whn[when], rot, na, best * 1e3 / launches,
               ld / grid, at / grid, bad ? "  SUMS WRONG" : "");
        return 0;
    };
    if (run(0, 0, 0, naddr, 0)) return 1;
    for (int op = 1; op <= 6; op++) {
        const int na = op == 6 ? naddr / 2 : naddr;
        for (int when = 0; when <= 2; when++)
            if (run(op, when, 5, na, 1)) return 1;
    }
    if (run(1, 2, 0, naddr, 0)) return 1;      // lock step
    if (run(1, 2, 1, naddr, 0)) return 1;
    if (run(1, 2, 9, naddr, 0)) return 1;
    if (run(1, 1, 5, naddr, 4)) return 1;      // more thinly spread
    // the reader: 256 workgroups each read the accumulator the atomics have just written
    {
        unsigned long long *out; CHK(hipMalloc(&out, grid * 8));
        a.op = 1; a.when = 2; a.rot = 5; a.naddr = naddr; a.src = src;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipStreamSynchronize(st));
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; i++) { a.src = src + (size_t)(i % NB) * wb; hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, st, a); hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), 0, st, a.acc64, naddr, out); }
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("u64 add behind the stream + a reader kernel (every workgroup reads all %d sums): %7.2f us per pair\n", naddr, best * 1e3 / launches);
        a.op = 0;
        best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipStreamSynchronize(st));
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; i++) { a.src = src + (size_t)(i % NB) * wb; hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, st, a); hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), 0, st, a.acc64, naddr, out); }
            CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("no atomics + the same reader kernel:                                              %7.2f us per pair\n", best * 1e3 / launches);
    }
    return 0;
}
