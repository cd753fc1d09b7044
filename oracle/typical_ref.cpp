// typical_ref.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// extern "C" shim over the reference's OWN sampler, include/rwkv/sampler/typical.h:20-66 (NumCpp), compiled from where
// it lies under /root/reference by oracle/Makefile into oracle/_ref/libtypical_ref.so (plain g++: NumCpp is header-only).
// tools/make_typical_golden.py draws from it to make tests/golden/typical_ref.npz; tests/test_sampler_ref_cpu.py
// pins include/rwkv_sampler.h (and through it the device sampler, tests/test_sampler_gpu.py) against it.
#include <cstdint>
#include "rwkv/sampler/typical.h"

extern "C" {

void typical_ref_seed(uint32_t seed) { nc::random::seed(seed); }

// n independent draws of typical(logits, temp, tau) (typical.h:20-58) from the same logits vector
void typical_ref_draw(const float *logits, float temp, float tau, int n, int *out)
{
    std::vector<float> l(logits, logits + 50277);
    for (int i = 0; i < n; i++) out[i] = typical(l.data(), temp, tau);
}

} // extern "C"
