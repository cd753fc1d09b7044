// greedy_app.cpp -- a reference-style caller (shape of examples/storygen/storygen.cpp:29-73 with argmax in
// place of typical()) written against this repo's drop-in include/rwkv.h.  Used by tests/test_cpp_api.py.
//   greedy_app <model.bin> <first_token> <n>   -> prints the n picked ids, then "state_ok" after a snapshot/restore check
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "rwkv.h"
#include "rwkv_sampler.h"

static unsigned long long pick(float *out)
{
    out[0] = -99;   // storygen.cpp:66
    unsigned long long best = 0;
    for (unsigned long long i = 1; i < 50277; i++) if (out[i] > out[best]) best = i;
    return best;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    RWKV model;
    try { model.forward(1); return 3; } catch (const std::runtime_error &e) { if (strcmp(e.what(), "RWKV not loaded")) return 4; }
    model.loadFile(argv[1], 2);
    try { model.loadFile(argv[1]); return 5; } catch (const std::runtime_error &e) { if (strcmp(e.what(), "RWKV already loaded")) return 6; }
    try { model.forward(std::vector<unsigned long long>{1, 2, 3}, GPT); return 7; } catch (const std::runtime_error &) {}
    unsigned long long tk = strtoull(argv[2], nullptr, 10);
    const int n = atoi(argv[3]);
    long long last = model.loadContext(std::vector<long long>{5, 6, 7});   // chunks of maxContext = 2 tokens
    if (last != 7) return 8;
    RWKVState snap = model.state->getSubState(0);                           // storygen.cpp:31
    for (int i = 0; i < n; i++) { tk = pick(model.forward(tk)); printf("%llu ", tk); }
    printf("\n");
    // restore the snapshot and replay: identical ids (RWKVState is a value type, rwkv.h:140-242)
    model.state->setSubState(snap, 0);
    unsigned long long tk2 = strtoull(argv[2], nullptr, 10), ok = 1;
    model.residentState = true; model.pushState();
    std::vector<unsigned long long> ids = model.decodeGreedy(tk2, n);       // device-side loop must agree with the host loop
    tk2 = strtoull(argv[2], nullptr, 10);
    model.state->setSubState(snap, 0); model.pushState();
    for (int i = 0; i < n; i++) { tk2 = pick(model.forward(tk2)); ok &= (tk2 == ids[i]); }
    int s = typical(model.out, 0.9f, 0.8f);
    printf("%s %d\n", ok && s >= 0 && s < 50277 ? "state_ok" : "state_BAD", s);
    return 0;
}
