// rwkv_sampler.h -- typical sampling over the 50277 logits (host post-processing).
//
// Behavioural mirror of reference include/rwkv/sampler/typical.h:20-66, pinned to it by
// tests/test_sampler_ref_cpu.py (20 000 draws of the reference's own function per case,
// tests/golden/typical_ref.npz).  Written against <algorithm>/<random> only -- the vendored NumCpp
// the reference pulls in for this one function is not needed.
//
// What the reference COMPUTES differs from the python recipe quoted in its header comment (softmax,
// entropy H, sort by |-log p - H|, keep the smallest set whose cumulative probability reaches tau,
// p^(1/temp), draw) in two places, both visible in the 20 000-draw histograms:
//   * the cut `probs[shifted_logits > sorted_logits[cutoff]] = 0` (typical.h:50) assigns into a TEMPORARY:
//     NumCpp's NdArray::operator[](NdArray<bool>) returns a copy (NumCpp/NdArray/NdArrayCore.hpp:778-781),
//     so nothing is cut and tau has no effect;
//   * `nc::power(probs, 1.0 / _temp)` (typical.h:52) takes a uint8 exponent (NumCpp/Functions/power.hpp:68):
//     1/temp is truncated to an integer n -- temp 0.9 or 0.8 -> n = 1 (no temperature at all), temp 0.5 ->
//     n = 2, temp > 1 -> n = 0, i.e. every weight becomes 1 and the draw is uniform over all 50277 ids.
// So typical(logits, temp, tau) draws from softmax(logits)^n, n = uint8(1/temp).  A drop-in has to give the
// reference's results, so THAT is the default here (recipe = false); recipe = true (or
// -DRWKV_TYPICAL_RECIPE=1 for the plain typical() calls) gives what the comment documents.
// Off the hot path; kept because the pybind surface exposes it.
#ifndef RWKV_SAMPLER_H
#define RWKV_SAMPLER_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

// the generator behind typical(): seeded from the OS like nc::random's default, or from the environment variable
// RWKV_SAMPLER_SEED for reproducible runs
inline std::mt19937_64 &rwkv_sampler_rng()
{
    static std::mt19937_64 g{[] {
        const char *e = std::getenv("RWKV_SAMPLER_SEED");
        return e ? (unsigned long long)std::strtoull(e, nullptr, 10) : (unsigned long long)std::random_device{}();
    }()};
    return g;
}

#ifndef RWKV_TYPICAL_RECIPE
#define RWKV_TYPICAL_RECIPE 0
#endif

// weights of the sampling distribution (unnormalised): the part of typical() before the draw
inline std::vector<double> typical_weights(const float *_logits, float _temp, float _tau, bool recipe = RWKV_TYPICAL_RECIPE != 0)
{
    const int len = 50277;
    std::vector<double> probs(len), shifted(len);
    double mx = _logits[0];
    for (int i = 1; i < len; i++) mx = std::max<double>(mx, _logits[i]);
    double z = 0;
    for (int i = 0; i < len; i++) { probs[i] = std::exp((double)_logits[i] - mx); z += probs[i]; }
    double ent = 0;
    for (int i = 0; i < len; i++) {
        probs[i] /= z;
        const double nl = -std::log(probs[i]);
        shifted[i] = nl;
        const double t = nl * probs[i];
        if (!std::isnan(t)) ent += t;
    }
    if (recipe) {
        for (int i = 0; i < len; i++) shifted[i] = std::fabs(shifted[i] - ent);
        std::vector<int> ids(len);
        std::iota(ids.begin(), ids.end(), 0);
        std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return shifted[a] < shifted[b]; });
        double cum = 0;
        int cutoff = 0;
        for (int i = 0; i < len; i++) { cum += probs[ids[i]]; if (cum < (double)_tau) cutoff++; }
        if (cutoff >= len) cutoff = len - 1;
        const double thr = shifted[ids[cutoff]];
        for (int i = 0; i < len; i++) if (shifted[i] > thr) probs[i] = 0;
    }
    if (_temp != 1.0f) {
        if (recipe) {
            for (int i = 0; i < len; i++) probs[i] = std::pow(probs[i], 1.0 / (double)_temp);
        } else {      // nc::power(NdArray<double>, uint8): integer exponent, repeated multiplication (NumCpp/Utils/power.hpp:46-60)
            const double e = 1.0 / (double)_temp;
            const unsigned n = e >= 255.0 ? 255u : (unsigned)(unsigned char)e;
            for (int i = 0; i < len; i++) {
                double r = n == 0 ? 1.0 : probs[i];
                for (unsigned k = 1; k < n; k++) r *= probs[i];
                probs[i] = r;
            }
        }
    }
    return probs;
}

// deterministic draw for a given uniform u in [0, 1): inverse CDF in token order -- the draw the device
// sampler (csrc/sampler.hip.h, rwkv_sample_typical) makes, so the two can be compared token for token
inline int typical_u(const float *_logits, float _temp, float _tau, double u, bool recipe = RWKV_TYPICAL_RECIPE != 0)
{
    const std::vector<double> w = typical_weights(_logits, _temp, _tau, recipe);
    double total = 0;
    for (double v : w) total += v;
    const double target = u * total;
    double c = 0;
    int last = 0;
    for (int i = 0; i < (int)w.size(); i++) {
        if (w[i] > 0) { c += w[i]; last = i; if (target < c) return i; }
    }
    return last;
}

inline int typical(float *_logits, float _temp = 0.9, float _tau = 0.8)
{
    const std::vector<double> probs = typical_weights(_logits, _temp, _tau);
    std::discrete_distribution<int> d(probs.begin(), probs.end());
    return d(rwkv_sampler_rng());
}

inline std::vector<unsigned long long> typical(int batchsize, float *_logits, float _temp = 0.9, float _tau = 0.8)
{
    std::vector<unsigned long long> out;
    for (int i = 0; i < batchsize; i++) out.push_back(typical(&_logits[i * 50277], _temp, _tau));
    return out;
}

#endif
