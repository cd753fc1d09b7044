// host sampler check: reads 50277 float logits from a file, prints typical_u() for each u on the command line
// (environment RWKV_APP_TRUNCATE=1: the documented typical cut instead of the reference's as-compiled behaviour)
// usage: sampler_app logits.bin temp tau u0 [u1 ...]
//        sampler_app logits.bin temp tau --weights out.f64   (dumps typical_weights(): 50277 doubles)
//        sampler_app logits.bin temp tau --draw n            (prints n draws of typical(), the randomised sampler)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rwkv_sampler.h"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    std::vector<float> l(50277);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(l.data(), sizeof(float), l.size(), f) != l.size()) return 3;
    fclose(f);
    const float temp = (float)atof(argv[2]), tau = (float)atof(argv[3]);
    const char *te = getenv("RWKV_APP_TRUNCATE");
    const bool trunc = te && te[0] == '1';
    if (std::string(argv[4]) == "--weights") {
        if (argc < 6) return 2;
        const std::vector<double> w = typical_weights(l.data(), temp, tau, trunc);
        FILE *o = fopen(argv[5], "wb");
        if (!o || fwrite(w.data(), sizeof(double), w.size(), o) != w.size()) return 5;
        fclose(o);
        return 0;
    }
    if (std::string(argv[4]) == "--draw") {
        if (argc < 6) return 2;
        for (int k = 0; k < atoi(argv[5]); k++) printf("%d\n", typical(l.data(), temp, tau));
        return 0;
    }
    for (int i = 4; i < argc; i++) printf("%d\n", typical_u(l.data(), temp, tau, atof(argv[i]), trunc));
    // the randomised draw stays inside the kept set
    const std::vector<double> w = typical_weights(l.data(), temp, tau);
    for (int k = 0; k < 8; k++) { const int t = typical(l.data(), temp, tau); if (!(w[t] > 0)) return 4; }
    return 0;
}
