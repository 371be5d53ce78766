// SAIPBSelfCTree.h -- hash-guided seed-to-seed extension as host code over the device's FM primitives.
// Interface and behaviour of the reference's SAIPBSelfCorrectTree (PacBio/SAIPBSelfCTree.h:140-287): addHashBySingleSeed
// (.cpp:704-787) collects the k-mers of the reads that overlap a seed by LF-walking the rows of its interval, and
// mergeTwoSeedsUsingHash (.cpp:91-256) extends the source seed base by base towards the target, keeping only extensions whose
// k-mer was seen near the same distance from the seed.  (The reference never instantiates the class: its call site is commented
// out, PacBioHybridCorrectionProcess.cpp:1074-1130; the parameters of that call site are what tests use.)
//
// Re-designed for a device back end: the tree does not touch the index itself.  Every step asks an FMAccess for (a) bi-intervals
// of whole k-mers, (b) one batch of Occ queries for the eight extensions of all live leaves, (c) LF-walks of up to 60 rows at
// once -- over the C ABI these are lrsc_find_kmers, lrsc_rank and lrsc_lf_walk, i.e. three kernel launches per step instead of
// hundreds of pointer-chasing calls.  K-mers are 2-bit packed keys (smallKmerSize <= 31), a leaf is its string + interval pair.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lrsc.h"

namespace stride {

class FMAccess {
public:
    virtual ~FMAccess() {}
    // findInterval(pRBWT, reverse(w)) and findInterval(pBWT, reverseComplement(w)) for n k-mers of equal length
    virtual void findBiIntervals(const std::vector<std::string>& kmers, std::vector<lrsc_biinterval>& out) = 0;
    // Occ(base, idx) on the given strand (RLBWT::getOcc), idx may be -1
    virtual void occ(const std::vector<lrsc_rank_query>& q, std::vector<uint64_t>& out) = 0;
    virtual uint64_t pc(int strand, char base) const = 0;       // RLBWT::getPC
    // the characters met by LF-walking from each row of `strand` (walk order), until a '$' row or max_steps characters
    virtual void lfWalks(int strand, const std::vector<uint64_t>& rows, uint32_t max_steps, std::vector<std::string>& out) = 0;
};

// FMAccess over the C ABI (one lrsc_ctx)
class LrscFMAccess : public FMAccess {
public:
    LrscFMAccess(lrsc_ctx* ctx, const lrsc_index* index);
    void findBiIntervals(const std::vector<std::string>& kmers, std::vector<lrsc_biinterval>& out) override;
    void occ(const std::vector<lrsc_rank_query>& q, std::vector<uint64_t>& out) override;
    uint64_t pc(int strand, char base) const override;
    void lfWalks(int strand, const std::vector<uint64_t>& rows, uint32_t max_steps, std::vector<std::string>& out) override;
private:
    lrsc_ctx* m_ctx;
    lrsc_index_info m_info;
};

class SAIPBSelfCorrectTree {
public:
    SAIPBSelfCorrectTree(FMAccess& fm, const std::string& rawSeq, size_t srcmaxLength, size_t min_SA_threshold = 2, int maxLeavesAllowed = 64);
    // returns the seed's k-mer frequency (the caller's repeat filter); skipRepeat: give up above 128
    size_t addHashBySingleSeed(const std::string& seedStr, size_t largeKmerSize, size_t smallKmerSize, size_t maxLength, bool skipRepeat,
                               int expectedLength = -1);
    // 1 = merged; -1 high error, -2 exceeded the search depth, -3 too many leaves, -4 / -5 gave up early
    int mergeTwoSeedsUsingHash(const std::string& src, const std::string& dest, std::string& mergedseq, size_t hashKmerSize, size_t maxLeaves,
                               size_t minLength, size_t maxLength, size_t expectedLength);
    size_t hashkmerfreqs(const std::string& fwdkmer, size_t kmerposition) const;
    size_t hashEntries() const { return m_hash.size(); }

private:
    struct Feature {                      // KmerFeatures: occurrences of a k-mer, bucketed by distance from the seed (35 per bucket)
        std::vector<long long> freq;
        double maxAvgFreq = 0;
    };
    struct Leaf {
        std::string seq;                  // the whole string from the root
        size_t kmerCount = 0;
        lrsc_biinterval iv{};
    };
    struct Ext { char base; lrsc_biinterval iv; };

    static bool pack(const std::string& s, size_t from, size_t len, uint64_t& key);
    void insertKmer(uint64_t key, long long pos, size_t maxLength);
    long long sumOfFreq(const Feature& f, long long pos) const;
    void collectAlong(const std::string& seedStr, const lrsc_interval& iv, int strand, size_t smallKmerSize, size_t maxLength, int expectedLength);
    void refine(size_t kmerSize);
    void extensionsOfLeaves(std::vector<std::vector<Ext>>& out, size_t cutoff);
    bool isExtensionValid(const std::string& fwdkmer, double currAvgFreq, size_t& kmerFreq, size_t bcount);
    void attemptToExtend(const std::vector<std::vector<Ext>>& exts, std::vector<Leaf>& next, size_t hashKmerSize);

    FMAccess& m_fm;
    const std::string m_rawSeq;
    const size_t m_maxLength;
    size_t m_minSAThreshold, m_maxLeavesAllowed;
    int m_currentLength = 0, m_seedLength = 0;
    std::vector<Leaf> m_leaves;
    lrsc_biinterval m_terminal{};
    std::unordered_map<uint64_t, Feature> m_hash;
    size_t m_hashKmerSize = 0;            // the small k-mer size of the collected k-mers (keys carry no length)
};

} // namespace stride
