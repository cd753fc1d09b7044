// rwkv.h -- C++ drop-in for the reference's host API (reference include/rwkv.h ->
// include/rwkv/rwkv/rwkv.h: enum MODE, class RWKVState :140-242, class RWKV :245-429),
// re-implemented as a thin header-only wrapper over the C-ABI of librwkv_mi355x.so
// (include/rwkv_mi355x.h).  Same public names, argument meaning and error behaviour, so the
// reference's callers (examples/*/*.cpp, bindings/pybind/c_binding.cpp) compile against it:
//
//     g++ -std=c++17 app.cpp -I<repo>/include -L<repo>/rwkv-cpp-accelerated_amd/csrc -lrwkv_mi355x
//
// Differences, all documented in INTEGRATION.md:
//   * `tensors[i]` holds the device pointer of slot i where the engine keeps that tensor in file layout (the f32/f64
//     vectors, the five state arrays, X, BUFFER2 = logits; rwkv_tensor_device) and NULL for the uint8 matrices, which
//     are re-tiled at load, and for pure scratch; getTensorSize()/getTensorTypes() answer from the format table.
//   * `residentState = true` keeps the state on the device between calls (the reference re-uploads
//     and re-downloads 5 x L x D doubles around every forward, rwkv.h:353,372); the default keeps
//     the reference's host-authoritative semantics.
//   * like the reference's umbrella include/rwkv.h:1-3 this header also brings in the sampler (`typical`,
//     include/rwkv_sampler.h) and the tokenizer.  The tokenizer (GPT-NeoX BPE) is outside this engine's
//     scope and is NOT re-implemented: when the reference's own include/rwkv/tokenizer/tokenizer.h is on
//     the include path (-I<reference>/include) it is pulled in as it is and `RWKV::loadTokenizer` /
//     `loadContext(std::string)` (rwkv.h:312-319,395-413) forward to it; without it those two members do
//     not exist and loadContext() takes token ids.  Define RWKV_NO_TOKENIZER to keep it out.
//   * decodeGreedy()/decodeTypical() run on the device state.  In the default host-authoritative mode they
//     upload `state` first and download it afterwards, so a following forward() continues the sequence.
#ifndef RWKV_H
#define RWKV_H

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rwkv_mi355x.h"
#include "rwkv_sampler.h"
#if !defined(RWKV_NO_TOKENIZER) && !defined(RWKV_HAVE_TOKENIZER) && defined(__has_include)
#if __has_include("rwkv/tokenizer/tokenizer.h")
#include "rwkv/tokenizer/tokenizer.h"   // the reference's GPT2Tokenizer, compiled from where it lies
#define RWKV_HAVE_TOKENIZER 1
#endif
#endif

enum MODE { PARRALEL, GPT };   // reference enums/enum.h:2-5

// tensor-slot indices of model.bin, reference enums/enum.h:7-55
enum {
    X, EMBED, LAYERNORMS, STATEXY, STATEAA, STATEBB, STATEPP, STATEDD, BUFFER1, BUFFER2, BUFFER3, BUFFER4,
    MIXK, MIXV, MIXR, KM, VM, RM, KR, VR, RR, O1, O2, O3, ATTOUT, ATTOUTR, ATTOUTO, FFNMIXK, FFNMIXV,
    FFNK, FFNV, FFNR, FFNKR, FFNVR, FFNRR, FFNKO, FFNVO, FFNRO, FFNKBUFFER, FFNVBUFFER, FFNRBUFFER,
    DECAY, BONUS, HEAD, HEADR, HEADO
};

namespace rwkv_detail {
inline unsigned long long tensor_size(unsigned long long i, unsigned long long a, unsigned long long b)
{   // reference rwkv.h:124-128
    const unsigned long long V = RWKV_VOCAB;
    const unsigned long long s[46] = {b, V * b, 4 * (a + 1) * b, a * b, a * b, a * b, a * b, a * b, b, V, b, b,
        a * b, a * b, a * b, a * b * b, a * b * b, a * b * b, a * b, a * b, a * b, a * b, a * b, a * b,
        a * b * b, a * b, a * b, a * b, a * b, a * b * b * 4, a * b * b * 4, a * b * b,
        a * b, a * b * 4, a * b, a * b, a * b * 4, a * b, b, b, b * 4, a * b, a * b, V * b, b, b};
    return s[i];
}
inline unsigned long long tensor_type(unsigned long long i)
{   // reference rwkv.h:84
    const unsigned long long t[46] = {8, 4, 8, 8, 8, 8, 8, 8, 8, 4, 4, 4, 8, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4,
                                      4, 1, 4, 4, 8, 8, 1, 1, 1, 4, 4, 4, 4, 4, 4, 8, 8, 4, 8, 8, 1, 4, 4};
    return t[i];
}
inline void check(int rc)
{
    if (rc != RWKV_OK) throw std::runtime_error(rwkv_last_error());
}
} // namespace rwkv_detail

// reference rwkv.h:140-242 -- five host arrays [stateSize][num_layers][num_embed] doubles, value type
class RWKVState
{
public:
    double *statexy, *stateaa, *statebb, *statepp, *statedd;
    unsigned long long num_layers, num_embed, stateSize;

    RWKVState(unsigned long long num_layers, unsigned long long num_embed, unsigned long long stateSize)
        : num_layers(num_layers), num_embed(num_embed), stateSize(stateSize)
    {
        alloc();
        for (unsigned long long i = 0; i < total(); i++) statexy[i] = stateaa[i] = statebb[i] = statepp[i] = statedd[i] = 0;
    }
    RWKVState(const RWKVState &o) : num_layers(o.num_layers), num_embed(o.num_embed), stateSize(o.stateSize)
    {
        alloc();
        copy_from(o, 0, 0, total());
    }
    // sub-state `offset` of `o` as a 1-slot state (the reference ignores the L*D stride here,
    // rwkv.h:205-209 -- only offset 0 is ever used by its callers; this one honours it)
    RWKVState(const RWKVState &o, unsigned long long offset) : num_layers(o.num_layers), num_embed(o.num_embed), stateSize(1)
    {
        alloc();
        copy_from(o, offset * num_layers * num_embed, 0, num_layers * num_embed);
    }
    RWKVState &operator=(const RWKVState &o)
    {
        if (this != &o) {
            release();
            num_layers = o.num_layers; num_embed = o.num_embed; stateSize = o.stateSize;
            alloc();
            copy_from(o, 0, 0, total());
        }
        return *this;
    }
    ~RWKVState() { release(); }

    RWKVState getSubState(unsigned long long offset = 0)
    {
        if (offset >= stateSize)
            throw std::runtime_error("State get offset out of bounds, max offset is " + std::to_string(stateSize));
        return RWKVState(*this, offset);
    }
    void setSubState(RWKVState &other, unsigned long long offset = 0)
    {
        const unsigned long long n = num_layers * num_embed;
        for (unsigned long long i = 0; i < n; i++) {
            statexy[i + offset * n] = other.statexy[i]; stateaa[i + offset * n] = other.stateaa[i];
            statebb[i + offset * n] = other.statebb[i]; statepp[i + offset * n] = other.statepp[i];
            statedd[i + offset * n] = other.statedd[i];
        }
    }

private:
    unsigned long long total() const { return num_layers * num_embed * stateSize; }
    void alloc()
    {
        const unsigned long long n = total();
        statexy = new double[n]; stateaa = new double[n]; statebb = new double[n]; statepp = new double[n]; statedd = new double[n];
    }
    void release() { delete[] statexy; delete[] stateaa; delete[] statebb; delete[] statepp; delete[] statedd; }
    void copy_from(const RWKVState &o, unsigned long long src, unsigned long long dst, unsigned long long n)
    {
        for (unsigned long long i = 0; i < n; i++) {
            statexy[dst + i] = o.statexy[src + i]; stateaa[dst + i] = o.stateaa[src + i]; statebb[dst + i] = o.statebb[src + i];
            statepp[dst + i] = o.statepp[src + i]; statedd[dst + i] = o.statedd[src + i];
        }
    }
};

class GPT2Tokenizer;   // the reference's tokenizer (see header comment); only ever used through a pointer here

// reference rwkv.h:245-429
class RWKV
{
public:
    int **tensors = new int *[46]();    // reference rwkv.h:248; filled by loadFile (see header comment)
    unsigned long long num_layers = 0;
    unsigned long long num_embed = 0;
    float *out = nullptr;               // host logits [maxContext][50277]; forward() returns this
    unsigned long long maxContext = 1;
    RWKVState *state = nullptr;         // host state (authoritative unless residentState)
    GPT2Tokenizer *tokenizer = nullptr;
    bool ready = false;
    bool residentState = false;         // engine extension: keep state on the device between forwards
    // deprecated aliases of state->*, reference rwkv.h:270-274
    double *statexy = nullptr, *stateaa = nullptr, *statebb = nullptr, *statepp = nullptr, *statedd = nullptr;

    RWKV() {}
    explicit RWKV(int device) : device_(device) {}
    RWKV(const RWKV &) = delete;
    RWKV &operator=(const RWKV &) = delete;

    void loadFile(const std::string &filename, unsigned long long maxGPT = 1)
    {
        if (ready) throw std::runtime_error("RWKV already loaded");
        ensure_ctx();
        const int rc = rwkv_load_file(ctx_, filename.c_str(), maxGPT);
        if (rc == RWKV_E_IO) {   // the reference prints and exit(1)s (rwkv.cu:641-645); a library throws instead
            std::cout << rwkv_last_error() << std::endl;
            throw std::runtime_error(rwkv_last_error());
        }
        rwkv_detail::check(rc);
        num_layers = rwkv_n_layers(ctx_);
        num_embed = rwkv_n_embed(ctx_);
        for (int i = 0; i < 46; i++) tensors[i] = static_cast<int *>(rwkv_tensor_device(ctx_, i));
        std::cout << "n_layers: " << num_layers << std::endl << "n_embed: " << num_embed << std::endl;   // rwkv.cu:653-654
        state = new RWKVState(num_layers, num_embed, maxGPT);
        statexy = state->statexy; stateaa = state->stateaa; statebb = state->statebb; statepp = state->statepp; statedd = state->statedd;
        out = new float[(size_t)RWKV_VOCAB * maxGPT]();
        maxContext = maxGPT;
        ready = true;
    }

    unsigned long long getTensorSize(unsigned long long i) { return rwkv_detail::tensor_size(i, num_layers, num_embed); }
    unsigned long long getTensorTypes(unsigned long long i) { return rwkv_detail::tensor_type(i); }

    float *forward(std::vector<unsigned long long> token, MODE mode)
    {
        if (!ready) throw std::runtime_error("RWKV not loaded");
        if (token.size() > maxContext)
            throw std::runtime_error("Context too large, max context is " + std::to_string(maxContext));
        const uint64_t T = token.size();
        std::vector<uint64_t> t64(token.begin(), token.end());
        if (!residentState)   // setState, rwkv.h:353
            rwkv_detail::check(rwkv_set_state(ctx_, state->statexy, state->stateaa, state->statebb, state->statepp, state->statedd, T));
        rwkv_detail::check(rwkv_forward(ctx_, t64.data(), T, mode == PARRALEL ? RWKV_MODE_PARRALEL : RWKV_MODE_GPT));
        if (residentState)
            rwkv_detail::check(rwkv_get_output(ctx_, out, nullptr, nullptr, nullptr, nullptr, nullptr, T));
        else                  // getOutput, rwkv.h:372
            rwkv_detail::check(rwkv_get_output(ctx_, out, state->statexy, state->stateaa, state->statebb, state->statepp, state->statedd, T));
        return out;
    }
    float *forward(unsigned long long token) { return forward(std::vector<unsigned long long>{token}, GPT); }
    float *forward(std::vector<long long> token, MODE mode)
    {
        std::vector<unsigned long long> t2(token.begin(), token.end());
        return forward(t2, mode);
    }

    RWKVState emptyState() { return RWKVState(num_layers, num_embed, 1); }

    // prompt ingestion in chunks of maxContext tokens, GPT mode (reference rwkv.h:395-413);
    // returns the last prompt token like the reference does
    long long loadContext(const std::vector<long long> &initial, bool progress = false)
    {
        if (initial.empty()) throw std::runtime_error("empty context");
        for (size_t i = 0; i < initial.size(); i += maxContext) {
            const size_t e = std::min(i + (size_t)maxContext, initial.size());
            forward(std::vector<unsigned long long>(initial.begin() + i, initial.begin() + e), GPT);
            if (progress) { std::cout << "\r" << int(float(i) / initial.size() * 100) << "%"; std::flush(std::cout); }
        }
        return initial.back();
    }
#ifdef RWKV_HAVE_TOKENIZER
    // reference rwkv.h:312-319
    void loadTokenizer(std::string vocabPath)
    {
        auto _tokenizer = GPT2Tokenizer::load(vocabPath + "/vocab.json", vocabPath + "/merges.txt");
        if (!_tokenizer.has_value()) {
            std::cerr << "Failed to load tokenizer" << std::endl;
            return;
        }
        tokenizer = new GPT2Tokenizer(_tokenizer.value());
    }
    // reference rwkv.h:395-413 (prints "<first id>:token" like the reference does)
    long long loadContext(std::string input, bool progress = false)
    {
        if (!tokenizer) throw std::runtime_error("tokenizer not loaded");
        const std::vector<long long> initial = widen(tokenizer->encode(input));
        if (!initial.empty()) std::cout << initial[0] << ":token";
        return loadContext(initial, progress);
    }
#endif

    // engine extensions -------------------------------------------------------------------
    // explicit sync points for residentState mode
    void pushState() { rwkv_detail::check(rwkv_set_state(ctx_, state->statexy, state->stateaa, state->statebb, state->statepp, state->statedd, maxContext)); }
    void pullState() { rwkv_detail::check(rwkv_get_output(ctx_, nullptr, state->statexy, state->stateaa, state->statebb, state->statepp, state->statedd, maxContext)); }
    // device-side greedy continuation (storygen's loop with argmax; token 0 banned as out[0] = -99 does)
    std::vector<unsigned long long> decodeGreedy(unsigned long long first, unsigned long long n)
    {
        if (!ready) throw std::runtime_error("RWKV not loaded");
        std::vector<uint64_t> ids(n);
        if (!residentState) pushState();      // host state is authoritative: continue from it ...
        rwkv_detail::check(rwkv_decode_greedy(ctx_, first, n, ids.data()));
        if (!residentState) pullState();      // ... and leave it where the generated tokens ended
        return std::vector<unsigned long long>(ids.begin(), ids.end());
    }
    // typical sampling on the device from the logits of the last forward (no 201 KB download, no host sort);
    // u in [0, 1) is the caller's uniform, the draw is the inverse CDF in token order (rwkv_sampler.h typical_u)
    int sampleTypical(float temp, float tau, double u, bool ban0 = false, unsigned long long row = 0, bool recipe = RWKV_TYPICAL_RECIPE != 0)
    {
        if (!ready) throw std::runtime_error("RWKV not loaded");
        uint64_t tok = 0;
        rwkv_detail::check(rwkv_sample_typical(ctx_, row, temp, tau, u, (ban0 ? RWKV_SAMPLE_BAN0 : 0) | (recipe ? RWKV_SAMPLE_RECIPE : 0), &tok));
        return (int)tok;
    }
    // device-side sampled continuation: storygen's loop (examples/storygen/storygen.cpp:63-69) without host round trips
    std::vector<unsigned long long> decodeTypical(unsigned long long first, unsigned long long n, float temp = 0.9f, float tau = 0.8f,
                                                  unsigned long long seed = 0, bool recipe = RWKV_TYPICAL_RECIPE != 0)
    {
        if (!ready) throw std::runtime_error("RWKV not loaded");
        std::vector<uint64_t> ids(n);
        if (!residentState) pushState();
        rwkv_detail::check(rwkv_decode_typical(ctx_, first, n, temp, tau, seed, recipe ? RWKV_SAMPLE_RECIPE : 0, ids.data()));
        if (!residentState) pullState();
        return std::vector<unsigned long long>(ids.begin(), ids.end());
    }
    rwkv_ctx *handle() { return ctx_; }

    ~RWKV()
    {
        delete[] out;
        if (ctx_) rwkv_free(ctx_);
        delete[] tensors;
        delete state;
    }

private:
    rwkv_ctx *ctx_ = nullptr;
    int device_ = 0;
    template <typename T> static std::vector<long long> widen(const std::vector<T> &v) { return std::vector<long long>(v.begin(), v.end()); }
    void ensure_ctx()
    {
        if (!ctx_) rwkv_detail::check(rwkv_create(&ctx_, device_));
    }
};

#endif // RWKV_H
