// oracle/capi_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" surface of the CPU restatement, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never by the product.
#include "fm_oracle.hpp"
#include "probe_oracle.hpp"
#include "extend_oracle.hpp"
#include "dp_oracle.hpp"
#include "process_oracle.hpp"
#include "stdaln_oracle.hpp"
#include "saipb_oracle.hpp"
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
uint64_t orc_bwt_occ_calls(void* h) { (void)h; return RLBwt::occ_calls_tls(); }
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


// What --debugseed dumps beside the seeds: the seeds removeHitchhikingSeeds dropped (seed/error/<id>.seed, LongReadProbe.cpp:220-225;
// same 8-int records) and getSeqAttribute's ratio per position (extend/<id>.log, :170-171; NaN-free, 0 for reads shorter than k).
int64_t orc_find_seeds_debug(void* bwt, void* rbwt, const lrsc_params* p, const char* bases, const uint64_t* off, uint32_t n_reads,
                             uint32_t* outcast_count, int32_t* outcasts, uint64_t cap, float* ratio)
{
    KmerThreshold thr;
    thr.initialize(-1, 50, p->pb_coverage);
    const ProbeParameters pp = make_probe(static_cast<RLBwt*>(bwt), static_cast<RLBwt*>(rbwt), p, &thr);
    uint64_t total = 0;
    for(uint32_t r = 0; r < n_reads; ++r) {
        const std::string seq(bases + off[r], bases + off[r + 1]);
        KmerLog log;
        allocateKmerLog(log, pp.pool, seq.size());
        SeedFeature::SeedVector sv;
        ProbeDebug dbg;
        searchSeedsWithHybridKmers(pp, log, seq, sv, &dbg, nullptr);
        if(ratio) for(size_t i = 0; i < seq.size(); ++i) ratio[off[r] + i] = i < dbg.ratio.size() ? dbg.ratio[i] : 0.0f;
        outcast_count[r] = (uint32_t)dbg.outcast.size();
        for(const auto& s : dbg.outcast) {
            if(total >= cap) return -1;
            int32_t* o = outcasts + total * 8;
            o[0] = s.seedStartPos; o[1] = s.seedLen; o[2] = s.maxFixedMerFreq; o[3] = s.isRepeat ? 1 : 0;
            o[4] = s.startBestKmerSize; o[5] = s.endBestKmerSize; o[6] = s.startKmerFreq; o[7] = s.endKmerFreq;
            ++total;
        }
    }
    return (int64_t)total;
}

// KmerThreshold::initialize(-1, end, cov, "") as `stride kmerfreq` sets it up (kmerfreq.cpp:75): out[mode*(end+2)+k]
int orc_threshold_table_range(int cov, int end, float* out)
{
    KmerThreshold thr;
    thr.initialize(-1, end, cov);
    for(int m = 0; m < 3; ++m) for(int k = 0; k <= end + 1; ++k) out[m * (end + 2) + k] = thr.get(m, k);
    return 0;
}

// ---- stdaln global alignment with the PacBio matrix (Thirdparty/stdaln.c) -> { '|' count, score, path_len }
int orc_stdaln_global(const char* s1, const char* s2, int* out3)
{
    const StdalnGlobal r = stdaln_global_pacbio(s1, s2);
    out3[0] = r.matches; out3[1] = r.score; out3[2] = r.path_len;
    return 0;
}

// ---- SAIPBSelfCorrectTree (row f3), driven the way its one -- commented-out -- call site drives it
// (PacBioHybridCorrectionProcess.cpp:1083-1122): k-mers of three 17-mers of the source and of the reverse-complemented target are
// collected by LF-walks, then mergeTwoSeedsUsingHash(last 17-mer of the source, target, extendKmerSize 15).  `source` must be at
// least 51 characters.  Returns the FM-walk code (1, -1 .. -5); merged = source + pbseq.substr(17) when a path was found.
// stats6 = { steps, maxUsedLeaves, results, hash entries, sourceFreq of the last addHash, targetFreq }
int orc_saipb_merge(void* bwt, void* rbwt, const char* source_c, const char* between_c, const char* target_c, int dis, int max_leaves,
                    char* out, uint64_t cap, int64_t* stats6)
{
    const std::string source(source_c), between(between_c), target(target_c);
    if(source.size() < 51 || target.size() < 15) return -100;
    const double maxRatio = 1.1, minRatio = 0.9;
    const int minOffSet = 30;
    const size_t extendKmerSize = 15, srcKmerSize = 17;
    SaipbSelfCorrectTree tree(static_cast<RLBwt*>(bwt), static_cast<RLBwt*>(rbwt), between, 2);
    std::string srcStr = source.substr(source.length() - srcKmerSize);
    const size_t srcMaxLength = (size_t)(maxRatio * (dis + minOffSet) + srcStr.length() + extendKmerSize);
    size_t sourceFreq = 0;
    srcStr = source.substr(source.length() - srcKmerSize * 2, srcKmerSize);
    sourceFreq = tree.addHashBySingleSeed(srcStr, srcKmerSize, extendKmerSize, srcMaxLength, true);
    srcStr = source.substr(source.length() - srcKmerSize * 3, srcKmerSize);
    sourceFreq = tree.addHashBySingleSeed(srcStr, srcKmerSize, extendKmerSize, srcMaxLength, true);
    srcStr = source.substr(source.length() - (size_t)(srcKmerSize * 1.5), srcKmerSize);
    sourceFreq = tree.addHashBySingleSeed(srcStr, srcKmerSize, extendKmerSize, srcMaxLength, true);
    srcStr = source.substr(source.length() - srcKmerSize);
    const std::string rvcTargetStr = reverse_complement(target);
    const int targetMaxLength = (int)(maxRatio * (dis + minOffSet) + rvcTargetStr.length() + srcKmerSize);
    size_t expectedLength = (size_t)dis + rvcTargetStr.length();
    const size_t targetFreq = tree.addHashBySingleSeed(rvcTargetStr, srcKmerSize, extendKmerSize, (size_t)targetMaxLength, true, (int)expectedLength);
    int srcMinLength = (int)(minRatio * (dis - minOffSet) + srcStr.length() + extendKmerSize);
    if(srcMinLength < 0) srcMinLength = 0;
    expectedLength = srcStr.length() + (size_t)dis + target.length();
    std::string pbseq;
    const int rc = tree.mergeTwoSeedsUsingHash(srcStr, target, pbseq, extendKmerSize, (size_t)max_leaves, (size_t)srcMinLength, srcMaxLength, expectedLength);
    std::string merged;
    if(!pbseq.empty()) merged = source + pbseq.substr(srcKmerSize);
    if(out && cap > merged.size()) { std::memcpy(out, merged.data(), merged.size()); out[merged.size()] = 0; }
    if(stats6) {
        stats6[0] = (int64_t)tree.steps; stats6[1] = (int64_t)tree.maxUsedLeaves; stats6[2] = (int64_t)tree.numResults;
        stats6[3] = (int64_t)tree.hashSize(); stats6[4] = (int64_t)sourceFreq; stats6[5] = (int64_t)targetFreq;
    }
    return rc;
}

// ---- IntervalTree (PacBio/IntervalTree.cpp) -----------------------------------------------------
void* orc_itree_build(const uint64_t* start, const uint64_t* stop, const uint64_t* value, uint64_t n)
{
    std::vector<TreeInterval> v;
    v.reserve(n);
    for(uint64_t i = 0; i < n; ++i) v.emplace_back(start[i], stop[i], value[i]);
    IntervalTree* t = new IntervalTree();
    *t = IntervalTree(v);       // construct, then deep-copy assign (LongReadCorrectByOverlap.cpp:150-151)
    return t;
}
void orc_itree_free(void* h) { delete static_cast<IntervalTree*>(h); }
uint64_t orc_itree_query(void* h, uint64_t start, uint64_t stop, uint64_t* out_values, uint64_t cap)
{
    std::vector<TreeInterval> r;
    static_cast<IntervalTree*>(h)->findOverlapping(start, stop, r);
    for(uint64_t i = 0; i < r.size() && i < cap; ++i) out_values[i] = r[i].value;
    return r.size();
}

// ---- Overlapper::extendMatch ---------------------------------------------------------------------
int orc_extend_match(const char* s1, const char* s2, int start1, int start2, int bandwidth, int match, int gap,
                     int mismatch, int* out7, char* cigar, int cigar_cap)
{
    const SequenceOverlap ov = extend_match(s1, s2, start1, start2, bandwidth, match, gap, mismatch);
    out7[0] = ov.match[0].start; out7[1] = ov.match[0].end; out7[2] = ov.match[1].start; out7[3] = ov.match[1].end;
    out7[4] = ov.score; out7[5] = ov.edit_distance; out7[6] = ov.total_columns;
    std::strncpy(cigar, ov.cigar.c_str(), cigar_cap - 1);
    cigar[cigar_cap - 1] = 0;
    return (int)ov.cigar.size();
}

// ---- FM-extend: one seed-pair walk (LongReadSelfCorrectByOverlap) ---------------------------------------
static FMextendParameters make_fm(const RLBwt* bwt, const RLBwt* rbwt, const lrsc_params* p)
{
    FMextendParameters f;                          // PacBioSelfCorrection.cpp:208-215
    f.indices.bwt = bwt;
    f.indices.rbwt = rbwt;
    f.idmerLength = p->idmer_len;
    f.maxLeaves = p->max_leaves;
    f.minKmerLength = p->min_kmer_len;
    f.PBcoverage = (size_t)p->pb_coverage;
    f.ErrorRate = p->error_rate;
    return f;
}
// Returns extendOverlap's code (1, -1, -2, -3, -4); merged sequence into out (NUL-terminated).
int orc_extend_walk(void* bwt, void* rbwt, const lrsc_params* p, const char* src, const char* path, const char* trg,
                    int dis, int initk, int max_overlap, int min_sa_threshold, char* out, int out_cap, uint64_t* stats3)
{
    const FMextendParameters f = make_fm(static_cast<RLBwt*>(bwt), static_cast<RLBwt*>(rbwt), p);
    FMWalkResult2 r;
    LongReadSelfCorrectByOverlap tree(src, path, trg, dis, (size_t)initk, (size_t)max_overlap, f, (size_t)min_sa_threshold);
    const int code = tree.extendOverlap(r);
    if(stats3) { stats3[0] = tree.stats.steps; stats3[1] = tree.stats.leaf_expansions; stats3[2] = tree.stats.refine_calls; }
    if(code > 0) {
        if((int)r.mergedSeq.size() + 1 > out_cap) return -100;
        std::memcpy(out, r.mergedSeq.c_str(), r.mergedSeq.size() + 1);
    } else if(out_cap > 0)
        out[0] = 0;
    return code;
}

// ---- DP/MSA fallback: buildMultipleAlignment + calculateBaseConsensus -------------------------------------------
// Returns the number of rows of the alignment; consensus (NUL-terminated) into out; n3 = {retrieved strings fwd-seed,
// retrieved strings RC-seed, consensus length}.
int orc_dp_consensus(void* bwt, void* rbwt, const char* query, int k, int min_overlap, double min_identity, int coverage,
                     int min_call_coverage, char* out, int out_cap, int* n3)
{
    IndexSet idx;
    idx.bwt = static_cast<RLBwt*>(bwt);
    idx.rbwt = static_cast<RLBwt*>(rbwt);
    const std::string q(query);
    if(n3) {
        std::vector<std::string> a, b;
        const size_t maxLength = q.length() * 1.1 + 20;
        retrieveStr(q, (size_t)k, maxLength, idx, false, (size_t)coverage, a);
        retrieveStr(q, (size_t)k, maxLength, idx, true, (size_t)coverage, b);
        n3[0] = (int)a.size(); n3[1] = (int)b.size();
    }
    MultipleAlignment ma = buildMultipleAlignment(q, (size_t)k, (size_t)k, (size_t)min_overlap, min_identity, (size_t)coverage, idx);
    const std::string cons = ma.calculateBaseConsensus(min_call_coverage, -1);
    if(n3) n3[2] = (int)cons.size();
    if((int)cons.size() + 1 > out_cap) return -100;
    std::memcpy(out, cons.c_str(), cons.size() + 1);
    return (int)ma.getNumRows();
}

// ---- the whole per-read path (PacBioSelfCorrectionProcess::process + PostProcess) ---------------------------
struct OrcRun {
    std::string correct_fa, discard_fa, stats;
    std::vector<int64_t> counters;     // per read: 10 counters (PacBioSelfCorrectionResult order) + merge flag
    std::vector<int32_t> walks;        // flat: read, srcStart, trgStart, code, via
    std::vector<int32_t> walk_work;    // flat, same order: gap, extension steps, leaf expansions
    uint64_t walk_stats[3] = {0, 0, 0};
    uint64_t spec[6] = {0, 0, 0, 0, 0, 0};
};

void* orc_correct_reads(void* bwt, void* rbwt, const lrsc_params* p, const char* bases, const uint64_t* off,
                        uint32_t n_reads, const char* id_prefix)
{
    KmerThreshold thr;
    thr.initialize(-1, 50, p->pb_coverage);
    CorrectionParameters cp;
    cp.indices.bwt = static_cast<RLBwt*>(bwt);
    cp.indices.rbwt = static_cast<RLBwt*>(rbwt);
    cp.PBcoverage = p->pb_coverage;
    cp.ErrorRate = p->error_rate;
    cp.startKmerLen = p->start_kmer_len;
    cp.nextTarget = p->next_target;
    cp.maxLeaves = p->max_leaves;
    cp.idmerLen = p->idmer_len;
    cp.minKmerLen = p->min_kmer_len;
    cp.Split = p->split != 0;
    cp.NoDp = p->no_dp != 0;
    cp.FM_params = make_fm(cp.indices.bwt, cp.indices.rbwt, p);
    cp.probe = make_probe(cp.indices.bwt, cp.indices.rbwt, p, &thr);
    cp.pool = cp.probe.pool;

    SelfCorrectionProcess proc(cp);
    SelfCorrectionPostProcess post(cp.Split);
    OrcRun* run = new OrcRun();
    for(uint32_t r = 0; r < n_reads; ++r) {
        const std::string seq(bases + off[r], bases + off[r + 1]);
        const std::string id = std::string(id_prefix ? id_prefix : "r") + std::to_string(r);
        const CorrectionResult res = proc.process(id, seq);
        post.process(id, seq, res);
        const int64_t c[11] = {res.totalReadsLen, res.correctedLen, res.totalSeedNum, res.totalWalkNum, res.highErrorNum,
                               res.exceedDepthNum, res.exceedLeaveNum, res.FMNum, res.DPNum, res.seedDis, res.merge ? 1 : 0};
        run->counters.insert(run->counters.end(), c, c + 11);
        for(const auto& w : res.walks) {
            const int32_t v[5] = {(int32_t)r, w.srcStartPos, w.trgStartPos, w.code, w.via};
            run->walks.insert(run->walks.end(), v, v + 5);
            const int32_t x[3] = {w.gap, w.steps, w.leaf_expansions};
            run->walk_work.insert(run->walk_work.end(), x, x + 3);
        }
        run->walk_stats[0] += res.walk_stats.steps;
        run->walk_stats[1] += res.walk_stats.leaf_expansions;
        run->walk_stats[2] += res.walk_stats.refine_calls;
        for(int i = 0; i < 6; ++i) run->spec[i] += res.spec[i];
    }
    run->correct_fa = post.correct_fa;
    run->discard_fa = post.discard_fa;
    run->stats = post.stats_text();
    return run;
}
void orc_run_free(void* h) { delete static_cast<OrcRun*>(h); }
// which: 0 correct.fa, 1 discard.fa, 2 stats text
uint64_t orc_run_text(void* h, int which, char* out, uint64_t cap)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    const std::string& s = which == 0 ? r->correct_fa : which == 1 ? r->discard_fa : r->stats;
    if(out && cap >= s.size()) std::memcpy(out, s.data(), s.size());
    return s.size();
}
uint64_t orc_run_counters(void* h, int64_t* out, uint64_t cap)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    if(out && cap >= r->counters.size()) std::memcpy(out, r->counters.data(), r->counters.size() * 8);
    return r->counters.size();
}
uint64_t orc_run_walks(void* h, int32_t* out, uint64_t cap)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    if(out && cap >= r->walks.size()) std::memcpy(out, r->walks.data(), r->walks.size() * 4);
    return r->walks.size();
}
uint64_t orc_run_walk_work(void* h, int32_t* out, uint64_t cap)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    if(out && cap >= r->walk_work.size()) std::memcpy(out, r->walk_work.data(), r->walk_work.size() * 4);
    return r->walk_work.size();
}
void orc_run_walk_stats(void* h, uint64_t* out3)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    for(int i = 0; i < 3; ++i) out3[i] = r->walk_stats[i];
}
// source-k-mer prediction statistics of the run (process_oracle.hpp: CorrectionResult::spec)
void orc_run_spec_stats(void* h, uint64_t* out6)
{
    const OrcRun* r = static_cast<OrcRun*>(h);
    for(int i = 0; i < 6; ++i) out6[i] = r->spec[i];
}

} // extern "C"
