// oracle/saipb_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
// Restatement of SAIPBSelfCorrectTree's hash-guided seed-to-seed extension:
//   KmerFeatures                       PacBio/SAIPBSelfCTree.h:30-138
//   addHashBySingleSeed                PacBio/SAIPBSelfCTree.cpp:704-787      insertKmerToHash  :891-914
//   mergeTwoSeedsUsingHash             :91-256      initializeSearchTree / initializeTerminalIntervals  :52-88
//   attempToExtendUsingHash            :977-1111    isExtensionValid :1131-1175   hashkmerfreqs :1114-1129
//   refineSAInterval                   :1178-1187   getFMIndexRightExtensions :1213-1253   isTerminated :1258-1294
//   SAIntervalNode                     FMIndexWalk/SAINode.h:33-168, SAINode.cpp:7-104
// PARITY UNPINNED: the class is never instantiated in the reference (its one call site, PacBioHybridCorrectionProcess.cpp:1089,
// is inside a comment block), it needs google dense_hash to build, and the reference holds no output of it.  Pinned pieces
// underneath: findInterval / updateInterval / getChar / getOcc (fm_oracle) and aln_stdaln (stdaln_oracle).  The live debug
// prints of the reference (:132,984,997,1167,1172,1237) are not reproduced.
#pragma once
#include <cstdint>
#include <list>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "fm_oracle.hpp"

namespace lrsc_oracle {

class KmerFeatures {                       // SAIPBSelfCTree.h:30-138
public:
    KmerFeatures(long long pos, size_t maxIntervalSize, size_t intervalSize = 35);
    void add(long long pos);
    long long getTotalFreq() const { return m_totalFreq; }
    long long getSumOfFreq(long long pos) const;
    void setMaxAvgFreq(double f) { m_maxAvgFreq = f; }
    double getMaxAvgFreq() const { return m_maxAvgFreq; }
private:
    std::vector<long long> m_sumOfFreq, m_sumOfPos;
    long long m_intervalSize, m_totalFreq = 0, m_totalSum = 0;
    double m_maxAvgFreq = 0;
};

struct SaipbNode {                         // SAINode + SAIntervalNode
    std::string label;
    SaipbNode* parent = nullptr;
    std::vector<std::unique_ptr<SaipbNode>> children;
    size_t totalKmerCount = 0, lastKmerCount = 0;
    bool updated = false;
    Interval fwd, rvc;                     // fwdInterval (rBWT), rvcInterval (BWT)
    std::string getSuffix(size_t l) const;
    std::string getFullString() const;
    void addKmerCount(size_t c) { totalKmerCount += c; lastKmerCount = c; }
    SaipbNode* createChild(const std::string& ext);
};

struct SaipbResult { std::string thread; size_t SAICoverage; };

class SaipbSelfCorrectTree {
public:
    SaipbSelfCorrectTree(const RLBwt* bwt, const RLBwt* rbwt, const std::string& rawSeq, size_t srcmaxLength, size_t min_SA_threshold = 2,
                         int maxLeavesAllowed = 64);
    size_t addHashBySingleSeed(const std::string& seedStr, size_t largeKmerSize, size_t smallKmerSize, size_t maxLength, bool skipRepeat,
                               int expectedLength = -1);
    int mergeTwoSeedsUsingHash(const std::string& src, const std::string& dest, std::string& mergedseq, size_t hashKmerSize, size_t maxLeaves,
                               size_t minLength, size_t maxLength, size_t expectedLength);
    size_t hashkmerfreqs(const std::string& fwdkmer, size_t kmerposition) const;
    size_t hashSize() const { return kmerHash.size(); }
    // test visibility (not in the reference)
    size_t steps = 0, maxUsedLeaves = 0, numResults = 0;

private:
    void initializeSearchTree(const std::string& src, size_t hashKmerSize);
    void initializeTerminalIntervals(const std::string& dest, size_t hashKmerSize);
    void attempToExtendUsingHash(std::list<SaipbNode*>& newLeaves, size_t hashKmerSize, size_t minExtFreq);
    void refineSAInterval(size_t newKmer);
    struct Ext { char b; Interval fwd, rvc; };
    std::vector<Ext> getFMIndexRightExtensions(const SaipbNode* node, size_t IntervalSizeCutoff) const;
    void insertKmerToHash(const std::string& kmer, size_t seedStrLen, size_t currentLength, size_t smallKmerSize, size_t maxLength, int expectedLength);
    bool isExtensionValid(const std::string& fwdkmer, double& currAvgFreq, size_t& kmerFreq, size_t bcount);
    bool isTerminated(std::vector<SaipbResult>& results);

    const RLBwt* m_pBWT;
    const RLBwt* m_pRBWT;
    const std::string m_rawSeq;
    const size_t m_maxLength;
    size_t m_min_SA_threshold, m_maxLeavesAllowed;
    int m_expectedLength = 0, m_currentLength = 0, m_seedLength = 0;
    std::unique_ptr<SaipbNode> m_pRootNode;
    std::list<SaipbNode*> m_leaves;
    Interval m_fwdTerminatedInterval, m_rvcTerminatedInterval;
    std::unordered_map<std::string, KmerFeatures> kmerHash;        // DenseHashMap<std::string, KmerFeatures*> in the reference: look-ups only
    bool m_isLargeLeaveRemoved = false;
};

} // namespace lrsc_oracle
