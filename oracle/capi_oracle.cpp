// oracle/capi_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" surface of the CPU restatement, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never by the product.
#include "fm_oracle.hpp"
#include "probe_oracle.hpp"
#include "../include/lrsc.h"   // POD types only (lrsc_params, lrsc_biinterval); no product code is linked

#include <cstring>
#include <string>
#include <vector>

using namespace lrsc_oracle;

static thread_local std::string g_err;

static std::vector<std::string> split_reads(const char* bases, const uint64_t* off, uint64_t n)
{
    std::vector<std::string> reads;
    reads.reserve(n);
    for(uint64_t i = 0; i < n; ++i) reads.emplace_back(bases + off[i], bases + off[i + 1]);
    return reads;
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- FM-index ---------------------------------------------------------------------
void* orc_bwt_load(const char* path)
{
    RLBwt* b = new RLBwt();
    if(!b->load(path, &g_err)) { delete b; return nullptr; }
    return b;
}
void* orc_bwt_from_units(const uint8_t* units, uint64_t n_units, uint64_t num_strings, uint64_t num_symbols)
{
    RLBwt* b = new RLBwt();
    b->assign(std::vector<uint8_t>(units, units + n_units), num_strings, num_symbols);
    return b;
}
void orc_bwt_free(void* h) { delete static_cast<RLBwt*>(h); }
uint64_t orc_bwt_num_strings(void* h) { return static_cast<RLBwt*>(h)->num_strings(); }
uint64_t orc_bwt_num_symbols(void* h) { return static_cast<RLBwt*>(h)->num_symbols(); }
uint64_t orc_bwt_num_runs(void* h) { return static_cast<RLBwt*>(h)->num_runs(); }
uint64_t orc_bwt_pc(void* h, char b) { return static_cast<RLBwt*>(h)->pc(bwt_rank_of(b)); }
uint64_t orc_bwt_occ_calls(void* h) { return static_cast<RLBwt*>(h)->occ_calls; }
void orc_bwt_occ_batch(void* h, const char* b, const int64_t* idx, uint64_t n, uint64_t* out)
{
    const RLBwt* p = static_cast<RLBwt*>(h);
    for(uint64_t i = 0; i < n; ++i) out[i] = p->occ(bwt_rank_of(b[i]), idx[i]);
}
void orc_bwt_char_batch(void* h, const uint64_t* idx, uint64_t n, char* out)
{
    const RLBwt* p = static_cast<RLBwt*>(h);
    for(uint64_t i = 0; i < n; ++i) out[i] = p->get_char(idx[i]);
}
// decode into caller buffer of num_symbols bytes
void orc_bwt_decode(void* h, char* out)
{
    const std::string s = static_cast<RLBwt*>(h)->decode();
    std::memcpy(out, s.data(), s.size());
}
// k-mers are fixed-length, concatenated; out = n x {lower, upper}
void orc_find_intervals(void* h, const char* kmers, uint32_t k, uint64_t n, int64_t* out)
{
    const RLBwt* p = static_cast<RLBwt*>(h);
    for(uint64_t i = 0; i < n; ++i) {
        const Interval iv = p->find_interval(std::string(kmers + i * k, k));
        out[2 * i] = iv.lower;
        out[2 * i + 1] = iv.upper;
    }
}

// ---- index construction --------------------------------------------------------------
// Builds <out_path> (.bwt if reverse_reads==0, .rbwt otherwise) from reads given as
// concatenated bases + n+1 offsets.
int orc_build_bwt_file(const char* bases, const uint64_t* off, uint64_t n, int reverse_reads,
                       const char* out_path)
{
    const std::vector<std::string> reads = split_reads(bases, off, n);
    const std::string bwt = build_bwt_naive(reads, reverse_reads != 0);
    const std::vector<uint8_t> units = rl_encode(bwt);
    if(!write_bwt_file(out_path, n, bwt.size(), units)) { g_err = std::string("cannot write ") + out_path; return -1; }
    return 0;
}


// ---- KmerThreshold ------------------------------------------------------------------------
// out = 3 x 52 floats (k = 0..51), same shape as ref_threshold_table
int orc_threshold_table(int cov, float* out)
{
    KmerThreshold t;
    t.initialize(-1, 50, cov);
    for(int mode = 0; mode < 3; ++mode)
        for(int k = 0; k <= 51; ++k) out[mode * 52 + k] = t.get(mode, k);
    return 0;
}
int orc_threshold_text(int cov, char* out, int cap)
{
    KmerThreshold t;
    t.initialize(-1, 50, cov);
    const std::string s = t.table_text();
    if((int)s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// ---- LongReadProbe ---------------------------------------------------------------------------
static ProbeParameters make_probe(const RLBwt* bwt, const RLBwt* rbwt, const lrsc_params* p, const KmerThreshold* thr)
{
    ProbeParameters pp;
    pp.indices.bwt = bwt;
    pp.indices.rbwt = rbwt;
    pp.startKmerLen = p->start_kmer_len;
    pp.scanKmerLen = p->scan_kmer_len;
    pp.kmerLenUpBound = p->kmer_len_up_bound;
    pp.PBcoverage = p->pb_coverage;
    pp.mode = p->mode;
    pp.radius = p->radius;
    pp.hhRatio = p->hh_ratio;
    pp.offset = {{p->offset[0], p->offset[1], p->offset[2]}};
    pp.pool = {5, 9, p->scan_kmer_len};                         // PacBioSelfCorrection.cpp:108
    for(int i = 0; i < 3; ++i) pp.pool.insert(p->start_kmer_len + p->offset[i]);   // :204-205
    pp.Manual = p->manual != 0;
    pp.thresholds = thr;
    return pp;
}

// Same record layout as lrsc_kmer_grid: rec = (read_off[r] + pos) * n_k + slot.
int orc_kmer_grid(void* bwt, void* rbwt, const char* bases, const uint64_t* off, uint32_t n_reads,
                  const uint8_t* ks, uint32_t n_k, lrsc_biinterval* out_iv, uint8_t* out_size, uint8_t* out_count)
{
    IndexSet idx;
    idx.bwt = static_cast<RLBwt*>(bwt);
    idx.rbwt = static_cast<RLBwt*>(rbwt);
    for(uint32_t r = 0; r < n_reads; ++r) {
        const std::string seq(bases + off[r], bases + off[r + 1]);
        std::vector<KmerFeature> prev_row(n_k);
        for(size_t pos = 0; pos < seq.size(); ++pos) {
            const KmerFeature* prev = nullptr;
            for(uint32_t j = 0; j < n_k; ++j) {
                prev_row[j] = KmerFeature(idx, seq, pos, ks[j], prev);      // LongReadProbe.cpp:146-150
                prev = &prev_row[j];
                const uint64_t rec = (off[r] + pos) * n_k + j;
                const KmerFeature& f = prev_row[j];
                if(out_iv) {
                    out_iv[rec].fwd.lower = f.biInterval.fwd.lower; out_iv[rec].fwd.upper = f.biInterval.fwd.upper;
                    out_iv[rec].rvc.lower = f.biInterval.rvc.lower; out_iv[rec].rvc.upper = f.biInterval.rvc.upper;
                }
                if(out_size) out_size[rec] = (uint8_t)f.size;
                if(out_count) for(int c = 0; c < 4; ++c) out_count[rec * 4 + c] = (uint8_t)f.count[c];
            }
        }
    }
    return 0;
}

// Seeds of every read (LongReadProbe::searchSeedsWithHybridKmers).  Flat outputs:
//   seed_count[r]; per seed (in read order): 8 int32 =
//   {seedStartPos, seedLen, maxFixedMerFreq, isRepeat, startBestKmerSize, endBestKmerSize, startKmerFreq, endKmerFreq}
//   attribute (optional): one int8 per base (LongReadProbe::getSeqAttribute)
// Returns the total number of seeds, or -1 if seed_cap is too small.
int64_t orc_find_seeds(void* bwt, void* rbwt, const lrsc_params* p, const char* bases, const uint64_t* off,
                       uint32_t n_reads, uint32_t* seed_count, int32_t* seeds, uint64_t seed_cap, int8_t* attribute)
{
    KmerThreshold thr;
    thr.initialize(-1, 50, p->pb_coverage);                       // PacBioSelfCorrection.cpp:231
    const ProbeParameters pp = make_probe(static_cast<RLBwt*>(bwt), static_cast<RLBwt*>(rbwt), p, &thr);
    uint64_t total = 0;
    for(uint32_t r = 0; r < n_reads; ++r) {
        const std::string seq(bases + off[r], bases + off[r + 1]);
        KmerLog log;
        allocateKmerLog(log, pp.pool, seq.size());
        SeedFeature::SeedVector sv;
        std::vector<int> attr;
        searchSeedsWithHybridKmers(pp, log, seq, sv, nullptr, &attr);
        if(attribute) {
            if(attr.empty()) std::memset(attribute + off[r], 1, seq.size());   // read shorter than k: never computed
            else for(size_t i = 0; i < attr.size(); ++i) attribute[off[r] + i] = (int8_t)attr[i];
        }
        seed_count[r] = (uint32_t)sv.size();
        for(const auto& s : sv) {
            if(total >= seed_cap) return -1;
            int32_t* o = seeds + total * 8;
            o[0] = s.seedStartPos; o[1] = s.seedLen; o[2] = s.maxFixedMerFreq; o[3] = s.isRepeat ? 1 : 0;
            o[4] = s.startBestKmerSize; o[5] = s.endBestKmerSize; o[6] = s.startKmerFreq; o[7] = s.endKmerFreq;
            ++total;
        }
    }
    return (int64_t)total;
}

} // extern "C"
