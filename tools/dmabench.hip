// dmabench.hip -- what does ONE loader wave per CU sustain with global_load_lds_dwordx4 into an LDS ring, and what do the pieces of
// the ring loader protocol (round 2: built for the one-launch token, now the ring kernels of csrc/kernels.hip.h) cost?  256 workgroups x 512 threads, wave 7 loads, waves 0..6 consume.
//   mode 0: issue only (vmcnt cap), nobody reads the ring
//   mode 1: + vmcnt read (s_getreg IB_STS) and a `landed` store per unit
//   mode 2: + consumers: wait landed, copy a 4-unit group to registers, free it; loader waits for room through freeq flags (in order)
//   mode 3: mode 2 with consumers spinning without s_sleep
//   mode 4: mode 2, loader polls room with one LDS read per try and sleeps 4
//   mode 5: mode 2 + consumers do 16 dot4 per piece (stand-in for group_dot)
//   mode 6 / 7: mode 2 with a unit issued by ONE asm statement (one address register pair, M0 set once): instruction offsets on
//               both sides / M0 stepped on the LDS side.  Modes 2, 6, 7 verify every word that arrives in the ring.
// build: hipcc --offload-arch=gfx950 -O3 -o dmabench tools/dmabench.hip ; run: ./dmabench [MiB per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 4, NU = 26, R = 4, FQ = 32, NC = 7;

__device__ __forceinline__ void dma_piece(const uint8_t *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// one unit = S pieces, ONE address register pair, M0 set once: the instruction offset moves the global address AND the LDS address
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
// the same with M0 stepped instead of the instruction offset on the LDS side
__device__ __forceinline__ void dma_unit4m(const uint8_t *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned in_flight()
{
    const unsigned v = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 7);
    return (v & 0xfu) | ((v >> 18) & 0x30u);
}
struct Ctl { unsigned landed, pad[3]; unsigned freeq[FQ]; };

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const uint8_t *src, size_t per_wg, unsigned *sink, unsigned chunk_units)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctl *mc = reinterpret_cast<Ctl *>(smem);
    unsigned char *ring = smem + 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < sizeof(Ctl) / 4) reinterpret_cast<unsigned *>(mc)[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned nunits = (unsigned)(per_wg / (S * 1024)), ngroups = nunits / R;
    const uint8_t *base = src + (size_t)blockIdx.x * per_wg;
    if (wave == NC) {
        const unsigned ringa = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)ring;
        unsigned issued = 0, pos = 0, pub = 0, tail = 0;
        (void)base;
        for (unsigned g = 0; g < ngroups; g++) {
            if (MODE >= 2) {
                for (int it = 0; it < (1 << 20); it++) {
                    while (tail != g && __hip_atomic_load(&mc->freeq[tail % FQ], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == tail + 1u) tail++;
                    if (!(issued + R - tail * R > (unsigned)NU || g - tail >= (unsigned)FQ)) break;
                    const unsigned u = (issued * S - in_flight()) / S;
                    if ((int)(u - pub) > 0) { pub = u; if (lane == 0) __hip_atomic_store(&mc->landed, u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    __builtin_amdgcn_s_sleep(MODE == 4 ? 4 : 1);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                wait_vm<DEPTH - S>();
                if (MODE >= 1) {
                    const unsigned u = (issued * S - in_flight()) / S;
                    if ((int)(u - pub) > 0) { pub = u; if (lane == 0) __hip_atomic_store(&mc->landed, u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
                }
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ringa + pos * (unsigned)(S * 1024)));
                // unit u of this workgroup: chunk u / chunk_units of a chunk-major layout (every workgroup's chunk c lies in matrix c)
                const uint8_t *p = src + (((size_t)(issued / chunk_units) * gridDim.x + blockIdx.x) * chunk_units + issued % chunk_units) * (S * 1024) + lane * 16;
                if (MODE == 6) dma_unit4(p, dst);
                else if (MODE == 7) dma_unit4m(p, dst);
                else {
#pragma unroll
                    for (int s = 0; s < S; s++) dma_piece(p + s * 1024, dst + s * 1024);
                }
                issued++;
                pos = pos + 1 == (unsigned)NU ? 0u : pos + 1;
            }
        }
        wait_vm<0>();
        if (lane == 0) __hip_atomic_store(&mc->landed, issued, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    if (MODE < 2) return;
    unsigned acc = 0, bad = 0;
    for (unsigned kl = wave; kl < ngroups; kl += NC) {
        const unsigned uend = (kl + 1) * R;
        for (int it = 0; it < (1 << 20); it++) {
            if ((int)(__hip_atomic_load(&mc->landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - uend) >= 0) break;
            if (MODE != 3) __builtin_amdgcn_s_sleep(1);
        }
        u32x4 w[R][S];
        unsigned p0 = (uend - R) % (unsigned)NU;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(ring + (size_t)p0 * (S * 1024)) + lane;
#pragma unroll
            for (int s = 0; s < S; s++) w[r][s] = p[s * 64];
            p0 = p0 + 1 == (unsigned)NU ? 0u : p0 + 1;
        }
        if (lane == 0) __hip_atomic_store(&mc->freeq[kl % FQ], kl + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (MODE == 6 || MODE == 7 || MODE == 2) {
                    const unsigned unit = (uend - R) + r;
                    const unsigned word0 = (unsigned)(((((size_t)(unit / chunk_units) * gridDim.x + blockIdx.x) * chunk_units + unit % chunk_units) * (S * 1024) + s * 1024 + lane * 16) / 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) if (w[r][s][j] != word0 + j) bad++;
                } else if (MODE == 5) {
#pragma unroll
                    for (int j = 0; j < 12; j++) acc = __builtin_amdgcn_udot4(w[r][s][j & 3], 0x01020304u + j, acc, false);
                } else acc += w[r][s][0];
            }
    }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
    if (bad) atomicAdd(&sink[1024 + MODE], bad);
}

template <int MODE, int DEPTH>
int run(const uint8_t *src, size_t per_wg, unsigned *sink, int grid, const char *what, unsigned chunk_units = 0)
{
    const size_t smem = 1024 + (size_t)NU * S * 1024;
    if (!chunk_units) chunk_units = (unsigned)(per_wg / (S * 1024));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipEventRecord(e0));
        k<MODE, DEPTH><<<grid, 512, smem>>>(src, per_wg, sink, chunk_units);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)per_wg * grid;
    unsigned bad = 0;
    CHK(hipMemcpy(&bad, sink + 1024 + MODE, 4, hipMemcpyDeviceToHost));
    printf("mode %d depth %2d  %-46s %7.1f us  %6.2f TB/s  %5.1f KB/us per CU  mismatching words %u\n", MODE, DEPTH, what, best * 1e3, bytes / best / 1e9, per_wg / (best * 1e3) / 1024.0, bad);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 4;
    const size_t kib = argc > 2 ? atoi(argv[2]) : 0;      // second argument: KiB per workgroup instead (short kernels: what a launch-sized stream costs)
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const size_t per_wg = kib ? kib << 10 : mib << 20;
    uint8_t *src; unsigned *sink;
    CHK(hipMalloc(&src, per_wg * grid + (64u << 20))); CHK(hipMalloc(&sink, 8192));
    CHK(hipMemset(sink, 0, 8192));
    {   // every 32-bit word holds its own index
        const size_t tot = per_wg * grid + (64u << 20);
        unsigned *h = (unsigned *)malloc(tot);
        for (size_t i = 0; i < tot / 4; i++) h[i] = (unsigned)i;
        CHK(hipMemcpy(src, h, tot, hipMemcpyHostToDevice));
        free(h);
    }
    printf("%d workgroups x %zu KiB\n", grid, per_wg >> 10);
    if (run<0, 63>(src, per_wg, sink, grid, "issue only")) return 1;
    if (run<0, 32>(src, per_wg, sink, grid, "issue only")) return 1;
    if (run<0, 16>(src, per_wg, sink, grid, "issue only")) return 1;
    if (run<1, 63>(src, per_wg, sink, grid, "+ vmcnt read + landed store per unit")) return 1;
    if (run<2, 63>(src, per_wg, sink, grid, "+ consumers (copy, free), room via freeq")) return 1;
    if (run<2, 32>(src, per_wg, sink, grid, "+ consumers (copy, free), room via freeq")) return 1;
    if (run<3, 63>(src, per_wg, sink, grid, "consumers spin without s_sleep")) return 1;
    if (run<4, 63>(src, per_wg, sink, grid, "loader sleeps 4 between room polls")) return 1;
    if (run<5, 63>(src, per_wg, sink, grid, "consumers + 12 dot4 per 16 B")) return 1;
    if (run<6, 63>(src, per_wg, sink, grid, "one asm per unit, instruction offsets")) return 1;
    if (run<7, 63>(src, per_wg, sink, grid, "one asm per unit, M0 stepped")) return 1;
    if (run<6, 63>(src, per_wg, sink, grid, "mode 6, 320 KiB chunks (chunk-major layout)", 80)) return 1;
    if (run<6, 63>(src, per_wg, sink, grid, "mode 6, 192 KiB chunks", 48)) return 1;
    if (run<6, 63>(src, per_wg, sink, grid, "mode 6, 64 KiB chunks", 16)) return 1;
    if (run<6, 32>(src, per_wg, sink, grid, "mode 6, 192 KiB chunks", 48)) return 1;
    if (run<0, 63>(src, per_wg, sink, grid, "issue only, 192 KiB chunks", 48)) return 1;
    return 0;
}
