// rwkv_sampler.h -- typical sampling over the 50277 logits (host post-processing).
//
// Behavioural mirror of reference include/rwkv/sampler/typical.h:20-66 (which implements the
// python recipe quoted in its header comment with NumCpp): softmax, entropy, sort by
// |-log p - H|, keep the smallest set whose cumulative probability reaches tau, p^(1/temp),
// draw from the (unnormalised) discrete distribution.  Written against <algorithm>/<random>
// only -- the vendored NumCpp the reference pulls in for this one function is not needed.
// Off the hot path; kept because the pybind surface exposes it.
#ifndef RWKV_SAMPLER_H
#define RWKV_SAMPLER_H

#include <algorithm>
#include <cmath>
#include <numeric>
#include <random>
#include <vector>

inline std::mt19937_64 &rwkv_sampler_rng()
{
    static std::mt19937_64 g{std::random_device{}()};
    return g;
}

// weights of the typical-sampling distribution (unnormalised): the part of typical() before the draw
inline std::vector<double> typical_weights(const float *_logits, float _temp, float _tau)
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
    if (_temp != 1.0f) for (int i = 0; i < len; i++) probs[i] = std::pow(probs[i], 1.0 / (double)_temp);
    return probs;
}

// deterministic draw for a given uniform u in [0, 1): inverse CDF in token order -- the draw the device
// sampler (csrc/sampler.hip.h, rwkv_sample_typical) makes, so the two can be compared token for token
inline int typical_u(const float *_logits, float _temp, float _tau, double u)
{
    const std::vector<double> w = typical_weights(_logits, _temp, _tau);
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
