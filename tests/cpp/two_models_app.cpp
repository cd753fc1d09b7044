// two_models_app.cpp -- TEST program: two RWKV objects alive in one process (every RWKV owns its tensors[] table, reference
// rwkv.h:248,288; the pybind module hands out one per initRwkv, c_binding.cpp:28-33).  Compiled against BOTH boundary levels by
// oracle/Makefile: include/rwkv.h (two_models_l1) and the reference's own rwkv.h + integration/rwkv_backend_mi355x.cpp
// (two_models_l2).  Greedy decode interleaved between the two models, then one is destroyed and the other continues.
//   usage: two_models <modelA.bin> <modelB.bin> <n>     prints three lines of ids: A, B, B after A is gone
#include "rwkv.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static unsigned long long pick(const float *out)
{
    unsigned long long best = 1;
    for (unsigned long long i = 2; i < 50277; i++) if (out[i] > out[best]) best = i;      // out[0] banned (storygen.cpp:66)
    return best;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int n = atoi(argv[3]);
    RWKV *A = new RWKV();
    RWKV *B = new RWKV();
    A->loadFile(argv[1]);
    B->loadFile(argv[2], 2);
    unsigned long long ta = 11, tb = 11;
    std::vector<unsigned long long> ia, ib, ic;
    for (int i = 0; i < n; i++) {
        ta = pick(A->forward(ta)); ia.push_back(ta);
        tb = pick(B->forward(tb)); ib.push_back(tb);
    }
    delete A;                                             // freeTensors of A must not touch B
    for (int i = 0; i < n; i++) { tb = pick(B->forward(tb)); ic.push_back(tb); }
    for (auto v : ia) printf("%llu ", v); printf("\n");
    for (auto v : ib) printf("%llu ", v); printf("\n");
    for (auto v : ic) printf("%llu ", v); printf("\n");
    printf("layers %llu %llu\n", (unsigned long long)B->num_layers, (unsigned long long)B->num_embed);
    delete B;
    return 0;
}
