// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked by the product).
//
// A thin extern "C" driver over the *real* reference translation units that
// compile directly from /root/reference with plain g++ (no config.h, no
// google-sparsehash, no stand-in headers).  It is built by oracle/Makefile into
// oracle/_ref/liblrsc_ref.so and used by tests/ to pin the CPU restatement in
// oracle/*.cpp against the reference's own object code:
//
//   * RLBWT load / getOcc / getPC / getChar     SuffixTools/RLBWT.{h,cpp}
//   * index construction (ropebwt2, IO order)   SuffixTools/BWTCARopebwt.cpp:160-247
//   * KmerThreshold table                       PacBio/KmerThreshold.cpp:43-79
//   * IntervalTree build + findOverlapping      PacBio/IntervalTree.cpp:4-48,73-91
//   * Overlapper::extendMatch                   Thirdparty/overlapper.cpp:421-701
//   * BCode::load / BCode::validate             PacBio/BCode.cpp:27-153 (--onlyseed, kmercheck)
//   * KmerDistribution + compare()              Util/KmerDistribution.cpp:25-153 (kmercheck)
//   * aln_stdaln global alignment, PacBio matrix Thirdparty/stdaln.c:364-546,780-862 (SAIPBSelfCTree's result choice)
//
// Everything above BWTAlgorithms.h (findInterval, LongReadProbe, FM-extend,
// multiple_alignment) pulls Util/HashMap.h -> generated config.h + google
// sparsehash and is therefore unbuildable in this image; see DESIGN.md.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#include <sstream>

#include "RLBWT.h"
#include "BWTCARopebwt.h"
#include "KmerThreshold.h"
#include "IntervalTree.h"
#include "overlapper.h"
#include "BCode.h"
#include "KmerDistribution.h"
#include "stdaln.h"

extern "C" {

// ---- RLBWT -----------------------------------------------------------------
void* ref_bwt_load(const char* path)
{
    return new RLBWT(std::string(path), RLBWT::DEFAULT_SAMPLE_RATE_SMALL);
}
void ref_bwt_free(void* h) { delete static_cast<RLBWT*>(h); }
uint64_t ref_bwt_num_strings(void* h) { return static_cast<RLBWT*>(h)->getNumStrings(); }
uint64_t ref_bwt_num_symbols(void* h) { return static_cast<RLBWT*>(h)->getBWLen(); }
uint64_t ref_bwt_num_runs(void* h) { return static_cast<RLBWT*>(h)->getNumRuns(); }
uint64_t ref_bwt_pc(void* h, char b) { return static_cast<RLBWT*>(h)->getPC(b); }
// idx is passed signed so that -1 wraps exactly as `interval.lower - 1` does
// in BWTAlgorithms.h:70.
uint64_t ref_bwt_occ(void* h, char b, int64_t idx)
{
    return static_cast<RLBWT*>(h)->getOcc(b, (size_t)idx);
}
void ref_bwt_occ_batch(void* h, const char* b, const int64_t* idx, uint64_t n, uint64_t* out)
{
    const RLBWT* p = static_cast<RLBWT*>(h);
    for(uint64_t i = 0; i < n; ++i) out[i] = p->getOcc(b[i], (size_t)idx[i]);
}
char ref_bwt_char(void* h, uint64_t idx) { return static_cast<RLBWT*>(h)->getChar(idx); }
void ref_bwt_char_batch(void* h, const uint64_t* idx, uint64_t n, char* out)
{
    const RLBWT* p = static_cast<RLBWT*>(h);
    for(uint64_t i = 0; i < n; ++i) out[i] = p->getChar(idx[i]);
}

// ---- index build (stride index -a ropebwt2; StriDe/index.cpp:164-213) --------
// do_reverse=0 -> <prefix>.bwt (BWT of the reads), 1 -> .rbwt (BWT of reversed reads)
int ref_build_bwt(const char* fasta, const char* out, int threads, int do_reverse)
{
    BWTCA::runRopebwt2(std::string(fasta), std::string(out), threads, do_reverse != 0);
    return 0;
}

// ---- KmerThreshold -------------------------------------------------------------
// The reference object is a process-wide singleton that can be initialised once
// (KmerThreshold.cpp:45-46), so one process pins one coverage.  out = 3 x 52 floats.
int ref_threshold_table(int cov, float* out)
{
    KmerThreshold::Instance().initialize(-1, 50, cov, "");
    for(int mode = 0; mode < 3; ++mode)
        for(int k = 0; k <= 51; ++k)
            out[mode * 52 + k] = KmerThreshold::Instance().get(mode, k);
    return 0;
}

// ---- IntervalTree ----------------------------------------------------------------
void* ref_itree_build(const uint64_t* start, const uint64_t* stop, const uint64_t* value, uint64_t n)
{
    std::vector<TreeInterval<size_t> > v;
    v.reserve(n);
    for(uint64_t i = 0; i < n; ++i) v.emplace_back(start[i], stop[i], value[i]);
    IntervalTree<size_t>* t = new IntervalTree<size_t>();
    // same shape as LongReadCorrectByOverlap.cpp:150-151 (construct, then deep-copy assign)
    *t = IntervalTree<size_t>(v);
    return t;
}
void ref_itree_free(void* h) { delete static_cast<IntervalTree<size_t>*>(h); }
uint64_t ref_itree_query(void* h, uint64_t start, uint64_t stop, uint64_t* out_values, uint64_t cap)
{
    std::vector<TreeInterval<size_t> > r;
    static_cast<IntervalTree<size_t>*>(h)->findOverlapping(start, stop, r);
    for(uint64_t i = 0; i < r.size() && i < cap; ++i) out_values[i] = r[i].value;
    return r.size();
}

// ---- Overlapper::extendMatch -------------------------------------------------------
// Returns the fields the DP fallback consumes (LongReadOverlap.cpp:626-659).
int ref_extend_match(const char* s1, const char* s2, int start1, int start2, int bandwidth, int match, int gap,
                     int mismatch, int* out7, char* cigar, int cigar_cap)
{
    SequenceOverlap ov = Overlapper::extendMatch(std::string(s1), std::string(s2), start1, start2, bandwidth,
                                                 match, gap, mismatch);
    out7[0] = ov.match[0].start; out7[1] = ov.match[0].end; out7[2] = ov.match[1].start; out7[3] = ov.match[1].end;
    out7[4] = ov.score; out7[5] = ov.edit_distance; out7[6] = ov.total_columns;
    std::strncpy(cigar, ov.cigar.c_str(), cigar_cap - 1);
    cigar[cigar_cap - 1] = 0;
    return (int)ov.cigar.size();
}


// ---- BCode (PacBio/BCode.cpp) ------------------------------------------------------------------
// validate() on one block; -1 if the reference throws (std::map::at / substr on a malformed code).
int ref_bcode_validate(int pos, int ksize, int start, int end, const char* code, int rvc, const char* seq)
{
    try {
        const BCode block(start, end, std::string(code), rvc != 0);
        return BCode::validate(pos, ksize, block, std::string(seq)) ? 1 : 0;
    } catch(const std::exception&) {
        return -1;
    }
}
// load() may run once per process (BCode.cpp:29-33).  Dumps the log as "qname start end rvc code\n" lines in map order.
uint64_t ref_bcode_load_dump(const char* path, char* out, uint64_t cap)
{
    if(BCode::Log().empty()) BCode::load(std::string(path));
    std::ostringstream o;
    for(const auto& kv : BCode::Log())
        for(const BCode& b : kv.second)
            o << kv.first << ' ' << b.getStart() << ' ' << b.getEnd() << ' ' << (b.getRvc() ? 1 : 0) << ' ' << b.getCode() << '\n';
    const std::string s = o.str();
    if(out && cap >= s.size()) std::memcpy(out, s.data(), s.size());
    return s.size();
}

// ---- KmerDistribution (Util/KmerDistribution.cpp) ------------------------------------------------
// compare() over two frequency lists -> "total.box line" + "value.box line"
uint64_t ref_kd_compare(const int* crt, uint64_t n_crt, const int* err, uint64_t n_err, int cov, int ksize, char* out, uint64_t cap)
{
    KmerDistribution c, e;
    for(uint64_t i = 0; i < n_crt; ++i) c.add(crt[i]);
    for(uint64_t i = 0; i < n_err; ++i) e.add(err[i]);
    std::ostringstream t, v;
    compare(t, v, cov, ksize, c, e);
    const std::string s = t.str() + v.str();
    if(out && cap >= s.size()) std::memcpy(out, s.data(), s.size());
    return s.size();
}


// ---- stdaln (Thirdparty/stdaln.c): aln_stdaln(s1, s2, &aln_param_pacbio, ALN_TYPE_GLOBAL, 1) as SAIPBSelfCTree.cpp:186-194 calls it
// out3 = { number of '|' in outm, score, path_len }
int ref_stdaln_global(const char* s1, const char* s2, int* out3)
{
    AlnAln* aln = aln_stdaln(s1, s2, &aln_param_pacbio, 1, 1);
    int matches = 0;
    for(int i = 0; aln->outm[i] != '\0'; i++) if(aln->outm[i] == '|') matches++;
    out3[0] = matches; out3[1] = aln->score; out3[2] = aln->path_len;
    aln_free_AlnAln(aln);
    return 0;
}

} // extern "C"
