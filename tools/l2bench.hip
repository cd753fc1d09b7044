// l2bench.hip -- how fast can ONE CU take in data that sits in its XCD's L2?  (round 6)
// The chunk path's GEMMs are sliced over K because a workgroup that owned full K would have to pull the whole activation image (384 KiB
// per vector at 32 rows x 4096 inputs) through its CU, and every measurement so far says "a CU takes in ~25 KB/us whatever the source".
// This measures the source separately: every workgroup (one per CU, 512 threads) reads the SAME `region` KiB again and again -- from the second
// round on they are L2 hits -- `total` KiB per workgroup, (a) with global_load_lds_dwordx4 (LDS-DMA, what the GEMMs' image copy uses) and (b) with
// global_load_dwordx4 into registers, `depth` wave-instructions of 1 KiB in flight per wave, with `waves` of the 8 waves loading;
// (c) the same while the other waves stream private HBM data (the weights).  Prints KB/us per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_piece(const unsigned char *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// mode 0: LDS-DMA of the shared region; 1: register loads of the shared region; 2: mode 0 on waves < ws while waves >= ws stream private HBM (nt)
template <int DEPTH>
__global__ __launch_bounds__(512) void k_l2(const unsigned char *shared, size_t region, const unsigned char *priv, size_t priv_per_wg, int pieces, int ws, int mode,
                                            unsigned *sink, unsigned long long *t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)smem + (unsigned)wave * 16384u;
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    if (wave < ws) {
        const size_t npr = region >> 10;                      // 1 KiB pieces of the region
        size_t p = ((size_t)blockIdx.x * 37 + (size_t)wave * 11) % npr;      // every workgroup / wave starts somewhere else
        if (mode == 1) {
            u32x4 v[DEPTH];
            for (int i = 0; i < pieces; i += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) { v[d] = *reinterpret_cast<const u32x4 *>(shared + (p << 10) + lane * 16); p = p + 1 == npr ? 0 : p + 1; }
#pragma unroll
                for (int d = 0; d < DEPTH; d++) acc += v[d][0] ^ v[d][3];
            }
        } else {
            for (int i = 0; i < pieces; i++) {
                wait_vm<DEPTH - 1>();
                dma_piece(shared + (p << 10) + lane * 16, lds0 + (unsigned)(i & 15) * 1024u);
                p = p + 1 == npr ? 0 : p + 1;
            }
            wait_vm<0>();
        }
    } else if (mode == 2) {
        const unsigned char *q = priv + (size_t)blockIdx.x * priv_per_wg + (size_t)(wave - ws) * 1024;
        const int np = (int)(priv_per_wg >> 10), step = 8 - ws;
        for (int i = wave - ws; i < np; i += step) {
            wait_vm<DEPTH - 1>();
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep)
                         : "v"(priv + (size_t)blockIdx.x * priv_per_wg + (size_t)i * 1024 + lane * 16), "s"(lds0 + (unsigned)(i & 15) * 1024u) : "memory");
        }
        (void)q;
        wait_vm<0>();
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) { t[((size_t)blockIdx.x * 8 + wave) * 2] = t0; t[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = t1; }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t MB = 1 << 20;
    unsigned char *shared, *priv; unsigned *sink; unsigned long long *t;
    CHK(hipMalloc(&shared, 64 * MB)); CHK(hipMemset(shared, 1, 64 * MB));
    CHK(hipMalloc(&priv, 1024 * MB)); CHK(hipMemset(priv, 2, 1024 * MB));
    CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&t, 256 * 8 * 2 * 8));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_l2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_l2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
    std::vector<unsigned long long> h(256 * 8 * 2);
    auto run = [&](int depth, size_t region_kb, int pieces, int ws, int mode, size_t priv_kb) {
        double best = 1e30, bestp = 0;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipMemset(t, 0, h.size() * 8));
            if (depth == 16) k_l2<16><<<256, 512, 8 * 16384>>>(shared, region_kb << 10, priv, priv_kb << 10, pieces, ws, mode, sink, t);
            else k_l2<4><<<256, 512, 8 * 16384>>>(shared, region_kb << 10, priv, priv_kb << 10, pieces, ws, mode, sink, t);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0, sump = 0; int n = 0, np = 0;
            for (int b = 0; b < 256; b++) {
                unsigned long long s0 = ~0ull, s1 = 0, p0 = ~0ull, p1 = 0;
                for (int w = 0; w < 8; w++) {
                    const unsigned long long a = h[(b * 8 + w) * 2], e = h[(b * 8 + w) * 2 + 1];
                    if (w < ws) { s0 = a < s0 ? a : s0; s1 = e > s1 ? e : s1; } else if (mode == 2) { p0 = a < p0 ? a : p0; p1 = e > p1 ? e : p1; }
                }
                sum += (double)(s1 - s0) / 100.0; n++;
                if (mode == 2) { sump += (double)(p1 - p0) / 100.0; np++; }
            }
            const double us = sum / n;
            if (us < best) { best = us; bestp = np ? sump / np : 0; }
        }
        const double kb = (double)ws * pieces;
        printf("mode %d depth %2d region %5zu KiB waves %d: %.1f KiB per CU in %.2f us = %.1f KB/us per CU (%.2f TB/s chip)", mode, depth, region_kb, ws, kb, best, kb * 1.024 / best, kb * 1.024 / best * 256 / 1e6);
        if (mode == 2) printf("  | private stream %zu KiB in %.2f us = %.1f KB/us", priv_kb, bestp, priv_kb * 1.024 / bestp);
        printf("\n");
    };
    for (size_t region : {48, 384, 768, 3072, 24576})
        for (int mode : {0, 1})
            for (int ws : {8, 4, 2})
                run(16, region, 2048 / ws, ws, mode, 0);
    run(4, 384, 256, 8, 0, 0);
    run(4, 384, 256, 8, 1, 0);
    // the image from L2 on 1 / 2 / 4 waves while the other waves stream 320 KiB of private HBM data (a GEMM workgroup's weights)
    for (int ws : {1, 2, 4}) run(16, 384, 384 / ws, ws, 2, 320);
    for (int ws : {1, 2, 4}) run(16, 768, 768 / ws, ws, 2, 320);
    return 0;
}
