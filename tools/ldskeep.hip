// ldskeep -- does the content of a CU's LDS survive a kernel boundary on gfx950, and is "one whole-LDS workgroup per CU" a
// bijection launch after launch?  (The question behind carrying the first rows of kernel N + 1's weights across the boundary in
// LDS -- DESIGN.md 6.)  Kernel A fills its workgroup's whole LDS allocation with a pattern keyed by the CU it runs on; kernel B,
// launched behind it, counts how many words it finds intact for ITS CU.  Variants: nothing in between / a small-LDS kernel in
// between / a memset in between / the pair captured in a hipGraph.
//   hipcc -O2 --offload-arch=gfx950 tools/ldskeep.hip -o tools/ldskeep && tools/ldskeep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned cu_key()
{
    const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    return ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);        // xcc | se_id, sh_id, cu_id
}
__device__ __forceinline__ unsigned pat(unsigned key, unsigned i, unsigned salt) { return (key * 2654435761u) ^ (i * 40503u + salt); }

__global__ __launch_bounds__(512) void k_fill(unsigned *ids, int words, unsigned salt)
{
    extern __shared__ unsigned lds[];
    const unsigned key = cu_key();
    for (int i = threadIdx.x; i < words; i += 512) lds[i] = pat(key, (unsigned)i, salt);
    if (threadIdx.x == 0) ids[blockIdx.x] = key;
}
__global__ __launch_bounds__(512) void k_check(unsigned *ids, unsigned *intact, unsigned *first_bad, int words, unsigned salt)
{
    extern __shared__ unsigned lds[];
    const unsigned key = cu_key();
    unsigned c = 0, f = 0xffffffffu;
    for (int i = threadIdx.x; i < words; i += 512) { if (lds[i] == pat(key, (unsigned)i, salt)) c++; else if ((unsigned)i < f) f = (unsigned)i; }
    __syncthreads();
    unsigned &cnt = lds[words], &fb = lds[words + 1];          // behind the words that are checked
    if (threadIdx.x == 0) { cnt = 0u; fb = 0xffffffffu; }
    __syncthreads();
    atomicAdd(&cnt, c); atomicMin(&fb, f);
    __syncthreads();
    if (threadIdx.x == 0) { ids[blockIdx.x] = key; intact[blockIdx.x] = cnt; first_bad[blockIdx.x] = fb; }
}
__global__ __launch_bounds__(256) void k_small(unsigned *sink, int words)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0xdeadbeefu + (unsigned)i;
    __syncthreads();
    if (lds[(threadIdx.x * 7) % words] == 1u) sink[0] = 1u;
}

int main()
{
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const int lds_kb = 160, words = lds_kb * 1024 / 4 - 16;          // k_check keeps its two counters in the last words
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fill), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_check), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    unsigned *ida, *idb, *intact, *fbad, *sink, *scratch;
    CHK(hipMalloc(&ida, grid * 4)); CHK(hipMalloc(&idb, grid * 4)); CHK(hipMalloc(&intact, grid * 4)); CHK(hipMalloc(&fbad, grid * 4));
    CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&scratch, 64 << 20));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<unsigned> ha(grid), hb(grid), hi(grid), hf(grid);
    const char *names[] = {"A -> B", "A -> small-LDS kernel (256 x 16 KB) -> B", "A -> hipMemsetAsync 64 MB -> B", "A -> B as a hipGraph, replayed",
                           "A -> small-LDS kernel (1024 x 16 KB) -> B"};
    for (int variant = 0; variant < 5; variant++) {
        for (int rep = 0; rep < 3; rep++) {
            const unsigned salt = 1000u * variant + rep;
            auto body = [&](hipStream_t s) {
                hipLaunchKernelGGL(k_fill, dim3(grid), dim3(512), (size_t)lds_kb * 1024, s, ida, lds_kb * 1024 / 4, salt);
                if (variant == 1) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 16384, s, sink, 4096);
                if (variant == 4) hipLaunchKernelGGL(k_small, dim3(4 * grid), dim3(256), 16384, s, sink, 4096);
                if (variant == 2) CHK(hipMemsetAsync(scratch, 1, 64 << 20, s));
                hipLaunchKernelGGL(k_check, dim3(grid), dim3(512), (size_t)lds_kb * 1024, s, idb, intact, fbad, words, salt);
            };
            if (variant == 3) {
                hipGraph_t g; hipGraphExec_t ge;
                CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                body(st);
                CHK(hipStreamEndCapture(st, &g));
                CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int k = 0; k <= rep; k++) CHK(hipGraphLaunch(ge, st));
                CHK(hipStreamSynchronize(st));
                CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
            } else {
                body(st);
                CHK(hipGetLastError());
                CHK(hipStreamSynchronize(st));
            }
            CHK(hipMemcpy(ha.data(), ida, grid * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hb.data(), idb, grid * 4, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hi.data(), intact, grid * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hf.data(), fbad, grid * 4, hipMemcpyDeviceToHost));
            std::set<unsigned> sa(ha.begin(), ha.end()), sb(hb.begin(), hb.end());
            int same_cu = 0, full = 0; unsigned long long tot = 0; unsigned minfb = 0xffffffffu;
            for (int b = 0; b < grid; b++) {
                same_cu += ha[b] == hb[b]; full += hi[b] == (unsigned)(words); tot += hi[b];
                if (hf[b] < minfb) minfb = hf[b];
            }
            printf("%-46s rep %d: CUs used by A %zu, by B %zu, same set %d; block b on the same CU in A and B: %d / %d; workgroups that found "
                   "ALL %d words intact: %d / %d (%.2f %% of all words; lowest damaged word %u)\n",
                   names[variant], rep, sa.size(), sb.size(), (int)(sa == sb), same_cu, grid, words, full, grid,
                   100.0 * (double)tot / ((double)(words) * grid), minfb);
        }
    }
    return 0;
}
