// valubench.hip -- issue throughput of the candidate inner-loop instructions on gfx950 (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int OP> __global__ void k(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    const unsigned w = seed * 0x01020304u + 0x11223344u;
    const float x = 1.0001f;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) {   // 8 independent v_fma_f32
            f0 = fmaf(f0, x, 1.f); f1 = fmaf(f1, x, 1.f); f2 = fmaf(f2, x, 1.f); f3 = fmaf(f3, x, 1.f);
            f4 = fmaf(f4, x, 1.f); f5 = fmaf(f5, x, 1.f); f6 = fmaf(f6, x, 1.f); f7 = fmaf(f7, x, 1.f);
        } else if (OP == 1) {   // 8 v_dot4_u32_u8
            a0 = __builtin_amdgcn_udot4(w, a1, a0, false); a1 = __builtin_amdgcn_udot4(w, a2, a1, false);
            a2 = __builtin_amdgcn_udot4(w, a3, a2, false); a3 = __builtin_amdgcn_udot4(w, a4, a3, false);
            a4 = __builtin_amdgcn_udot4(w, a5, a4, false); a5 = __builtin_amdgcn_udot4(w, a6, a5, false);
            a6 = __builtin_amdgcn_udot4(w, a7, a6, false); a7 = __builtin_amdgcn_udot4(w, a0, a7, false);
        } else if (OP == 2) {   // 8 v_cvt_f32_ubyteN (+ xor to keep them live)
            f0 = (float)(a0 & 255u); f1 = (float)((a1 >> 8) & 255u); f2 = (float)((a2 >> 16) & 255u); f3 = (float)(a3 >> 24);
            f4 = (float)(a4 & 255u); f5 = (float)((a5 >> 8) & 255u); f6 = (float)((a6 >> 16) & 255u); f7 = (float)(a7 >> 24);
            asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 3) {   // 4 v_pk_fma_f32 (8 FMAs)
            f32x2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
            const f32x2 xx = {x, x}, one = {1.f, 1.f};
            p0 = __builtin_elementwise_fma(p0, xx, one); p1 = __builtin_elementwise_fma(p1, xx, one);
            p2 = __builtin_elementwise_fma(p2, xx, one); p3 = __builtin_elementwise_fma(p3, xx, one);
            f0 = p0[0]; f1 = p0[1]; f2 = p1[0]; f3 = p1[1]; f4 = p2[0]; f5 = p2[1]; f6 = p3[0]; f7 = p3[1];
        } else if (OP == 4) {   // cvt + fma pairs (the f32 GEMV inner step): 8 bytes
            f0 = fmaf((float)(a0 & 255u), x, f0); f1 = fmaf((float)((a0 >> 8) & 255u), x, f1);
            f2 = fmaf((float)((a0 >> 16) & 255u), x, f2); f3 = fmaf((float)(a0 >> 24), x, f3);
            f4 = fmaf((float)(a1 & 255u), x, f4); f5 = fmaf((float)((a1 >> 8) & 255u), x, f5);
            f6 = fmaf((float)((a1 >> 16) & 255u), x, f6); f7 = fmaf((float)(a1 >> 24), x, f7);
            asm volatile("" : "+v"(a0), "+v"(a1));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int OP> void run(const char *name, int ops_per_iter, unsigned *out, int threads)
{
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    k<OP><<<256, threads>>>(out, 1); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    k<OP><<<256, threads>>>(out, 2);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double waves_per_simd = threads / 64.0 / 4.0;
    const double instr = (double)ITERS * ops_per_iter * waves_per_simd;   // wave-instructions per SIMD
    printf("%-28s %4d thr/WG: %7.1f us -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, threads, ms * 1e3,
           ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
    unsigned *out; CHK(hipMalloc(&out, 256 * 1024 * 4));
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32 x8", 8, out, threads);
        run<1>("v_dot4_u32_u8 x8", 8, out, threads);
        run<2>("v_cvt_f32_ubyteN x8", 8, out, threads);
        run<3>("v_pk_fma_f32 x4 (8 fma)", 4, out, threads);
        run<4>("cvt+fma x8 (16 instr)", 16, out, threads);
    }
    return 0;
}
