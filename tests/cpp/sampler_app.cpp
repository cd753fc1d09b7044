// host sampler check: reads 50277 float logits from a file, prints typical_u() for each u on the command line
// usage: sampler_app logits.bin temp tau u0 [u1 ...]
#include <cstdio>
#include <cstdlib>
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
    for (int i = 4; i < argc; i++) printf("%d\n", typical_u(l.data(), temp, tau, atof(argv[i])));
    // the randomised draw stays inside the kept set
    const std::vector<double> w = typical_weights(l.data(), temp, tau);
    for (int k = 0; k < 8; k++) { const int t = typical(l.data(), temp, tau); if (!(w[t] > 0)) return 4; }
    return 0;
}
