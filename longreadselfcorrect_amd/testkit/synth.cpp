// synth.cpp -- deterministic synthetic genome / PacBio-like read generator.
//
// splitmix64 streams; read i is a pure function of (seed, i) so any rank can generate its own
// shard and the whole set is independent of how it is split.  Not part of the reference: the
// reference has no data generator; this only defines the benchmark/test workload.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/lrsc.h"
#include "lrsc_testkit.h"

namespace {

struct SplitMix {
    uint64_t s;
    explicit SplitMix(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

const char kBases[4] = {'A', 'C', 'G', 'T'};

inline char comp(char c)
{
    switch(c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; default: return 'A'; }
}

// one read into `out` (capacity cap); returns length or (uint64_t)-1 on overflow
uint64_t make_read(uint64_t seed, uint64_t read_id, const char* genome, uint64_t glen, uint32_t tmpl_len,
                   double p_del, double p_sub, double q_ins, char* out, uint64_t cap)
{
    SplitMix rng(seed ^ (0xD1B54A32D192ED03ull * (read_id + 1)));
    const uint64_t L = tmpl_len < glen ? tmpl_len : glen;
    const uint64_t start = rng.below(glen - L + 1);
    const bool rc = (rng.next() & 1) != 0;
    uint64_t n = 0;
    for(uint64_t t = 0; t < L; ++t) {
        char b = rc ? comp(genome[start + L - 1 - t]) : genome[start + t];
        const double u = rng.unit();
        if(u >= p_del) {
            if(u < p_del + p_sub) {
                // substitute with one of the three other bases
                const unsigned code = (b == 'A') ? 0 : (b == 'C') ? 1 : (b == 'G') ? 2 : 3;
                b = kBases[(code + 1 + rng.below(3)) & 3];
            }
            if(n >= cap) return (uint64_t)-1;
            out[n++] = b;
        }
        while(rng.unit() < q_ins) {
            if(n >= cap) return (uint64_t)-1;
            out[n++] = kBases[rng.below(4)];
        }
    }
    return n;
}

} // namespace

extern "C" int lrsc_synth_genome(uint64_t seed, uint64_t len, char* out)
{
    if(!out && len) return LRSC_ERR_ARG;
    SplitMix rng(seed);
    uint64_t i = 0;
    while(i < len) {
        uint64_t r = rng.next();
        for(int k = 0; k < 32 && i < len; ++k, r >>= 2) out[i++] = kBases[r & 3];
    }
    return LRSC_OK;
}

extern "C" int lrsc_synth_reads(uint64_t seed, const char* genome, uint64_t genome_len, uint64_t first_read,
                                uint32_t n_reads, uint32_t tmpl_len, double p_del, double p_sub, double p_ins,
                                char* out_bases, uint64_t cap, uint64_t* out_off)
{
    if(!genome || genome_len == 0 || !out_bases || !out_off || tmpl_len == 0) return LRSC_ERR_ARG;
    if(p_del < 0 || p_sub < 0 || p_ins < 0 || p_del + p_sub >= 1.0) return LRSC_ERR_ARG;
    const double q_ins = p_ins / (1.0 + p_ins);   // geometric: mean inserted bases per template base == p_ins
    // pass 1 (parallel): lengths; pass 2 (parallel): bases at their final offsets
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = std::max(1u, std::min(hw ? hw : 1u, 16u));
    std::vector<uint64_t> len(n_reads, 0);
    const uint64_t worst = (uint64_t)tmpl_len * 4 + 64;
    bool overflow = false;
    {
        std::vector<std::thread> th;
        for(unsigned t = 0; t < n_thr; ++t)
            th.emplace_back([&, t]() {
                std::vector<char> tmp(worst);
                for(uint32_t i = t; i < n_reads; i += n_thr) {
                    const uint64_t l = make_read(seed, first_read + i, genome, genome_len, tmpl_len, p_del, p_sub,
                                                 q_ins, tmp.data(), worst);
                    len[i] = (l == (uint64_t)-1) ? worst : l;
                }
            });
        for(auto& x : th) x.join();
    }
    out_off[0] = 0;
    for(uint32_t i = 0; i < n_reads; ++i) out_off[i + 1] = out_off[i] + len[i];
    if(out_off[n_reads] > cap) return LRSC_ERR_CAPACITY;
    {
        std::vector<std::thread> th;
        for(unsigned t = 0; t < n_thr; ++t)
            th.emplace_back([&, t]() {
                for(uint32_t i = t; i < n_reads; i += n_thr) {
                    const uint64_t l = make_read(seed, first_read + i, genome, genome_len, tmpl_len, p_del, p_sub,
                                                 q_ins, out_bases + out_off[i], len[i]);
                    if(l != len[i]) overflow = true;
                }
            });
        for(auto& x : th) x.join();
    }
    return overflow ? LRSC_ERR_CAPACITY : LRSC_OK;
}
