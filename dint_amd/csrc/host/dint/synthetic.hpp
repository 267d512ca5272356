// Deterministic synthetic posting lists shaped like web collections
// (SURVEY §8d configs 1, 2 and 4). There is no reference counterpart: the
// reference only reads collections from disk and its bundled test collection
// is absent (.MISSING_LARGE_BLOBS).
//
// * list lengths: power law with density ~ len^-alpha on [min_len, max_len]
//   (alpha = 1.5 gives ~98 % of the lists shorter than 4096 postings while
//   ~98 % of the postings sit in longer lists, which is the Gov2/ClueWeb shape);
// * d-gaps minus one, per list, from a two-state chain: a "cluster" state with
//   geometric gaps whose success probability grows with the list's density
//   (dense lists get long zero runs, as URL-ordered docIDs do) and a "sparse"
//   state whose mean is set so that the list spans about the whole universe.
//
// Every list is generated from its own counter-seeded stream, so any subset of
// lists can be produced independently and in parallel with identical results.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace dint {

struct splitmix64 {
    uint64_t s;
    explicit splitmix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    double unit() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
};

struct synth_params {
    uint64_t seed = 12345;
    uint32_t universe = 25000000;  // number of documents
    double alpha = 1.5;            // list-length power law exponent
    uint32_t min_len = 1;
    uint32_t max_len = 0;          // 0 -> universe / 3
    double stay_cluster = 0.90;    // P(cluster -> cluster)
    double stay_sparse = 0.75;     // P(sparse -> sparse)
    double p_cluster_min = 0.55;   // geometric p of the cluster state for sparse lists
    double p_cluster_max = 0.97;   // ... and for lists as dense as universe/3
};

// List lengths summing to >= target_postings (the last list is trimmed so the
// total is exact).
inline std::vector<uint32_t> synth_lengths(synth_params const& p, uint64_t target_postings) {
    uint32_t max_len = p.max_len ? p.max_len : std::max<uint32_t>(1, p.universe / 3);
    uint32_t min_len = std::max<uint32_t>(1, std::min(p.min_len, max_len));
    splitmix64 rng(p.seed ^ 0x5851f42d4c957f2dULL);
    std::vector<uint32_t> lens;
    uint64_t total = 0;
    double a = 1.0 - p.alpha;
    double lo = std::pow(double(min_len), a), hi = std::pow(double(max_len) + 1.0, a);
    while (total < target_postings) {
        double u = rng.unit();
        double x = std::fabs(a) < 1e-9 ? double(min_len) * std::pow((double(max_len) + 1.0) / double(min_len), u)
                                       : std::pow(lo + u * (hi - lo), 1.0 / a);
        uint64_t len = uint64_t(x);
        len = std::max<uint64_t>(min_len, std::min<uint64_t>(max_len, len));
        len = std::min<uint64_t>(len, target_postings - total);
        lens.push_back(uint32_t(len));
        total += len;
    }
    return lens;
}

inline uint32_t geometric0(splitmix64& rng, double log1mp) {
    // number of failures before the first success, success probability p
    double u = 1.0 - rng.unit();  // (0,1]
    double g = std::floor(std::log(u) / log1mp);
    return g >= 4294967295.0 ? 4294967295u : uint32_t(g);
}

// gaps[i] = docid[i] - docid[i-1] - 1 (docid[-1] = -1): what the encoders consume.
inline void synth_gaps(synth_params const& p, uint64_t list_id, uint32_t n, uint32_t* gaps) {
    splitmix64 rng(p.seed * 0x9e3779b97f4a7c15ULL + list_id * 0xd1342543de82ef95ULL + 1);
    double density = std::min(1.0, double(n) / double(p.universe));
    double w = std::min(1.0, std::sqrt(3.0 * density));
    double p_c = p.p_cluster_min + (p.p_cluster_max - p.p_cluster_min) * w;
    double mean_c = (1.0 - p_c) / p_c;
    double pi_c = (1.0 - p.stay_sparse) / ((1.0 - p.stay_cluster) + (1.0 - p.stay_sparse));
    double mean_all = 1.0 / density - 1.0;
    double mean_s = std::max(mean_c, (mean_all - pi_c * mean_c) / (1.0 - pi_c));
    double p_s = 1.0 / (1.0 + mean_s);
    double log_c = std::log1p(-p_c), log_s = std::log1p(-p_s);
    bool in_cluster = rng.unit() < pi_c;
    for (uint32_t i = 0; i != n; ++i) {
        gaps[i] = geometric0(rng, in_cluster ? log_c : log_s);
        double stay = in_cluster ? p.stay_cluster : p.stay_sparse;
        if (rng.unit() >= stay) in_cluster = !in_cluster;
    }
}

}  // namespace dint
