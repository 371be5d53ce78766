// oracle/extend_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the seed-to-seed FM-extend engine that `pbcorrect` actually calls:
// LongReadSelfCorrectByOverlap (PacBio/LongReadCorrectByOverlap.{h,cpp}), its node type
// SAIOverlapNode3 (FMIndexWalk/SAINode.{h,cpp}) and IntervalTree (PacBio/IntervalTree.{h,cpp}).
//
// Parity pin: IntervalTree is checked against the reference's own object code (oracle/_ref:
// PacBio/IntervalTree.cpp compiles directly) including its observable std::sort tie order.
// LongReadCorrectByOverlap.cpp / SAINode.cpp include Util/HashMap.h -> generated config.h +
// google sparsehash and cannot be built in this image: this engine is "parity unpinned" by a
// reference build; it is a line-by-line restatement (every function cites its source lines),
// with identical integer widths/signedness and double evaluation order.
#pragma once
#include <list>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "fm_oracle.hpp"

namespace lrsc_oracle {

// ---- PacBio/IntervalTree.h:9-71 ---------------------------------------------------------
struct TreeInterval {
    size_t start, stop, value;
    TreeInterval(size_t s, size_t e, size_t v) : start(s), stop(e), value(v) {}
};
class IntervalTree {
public:
    typedef std::vector<TreeInterval> intervalVector;
    IntervalTree() : center(0) {}
    IntervalTree(intervalVector& ivals, size_t depth = 16, size_t minbucket = 8, size_t leftextent = 0,
                 size_t rightextent = 0, size_t maxbucket = 512);                       // IntervalTree.cpp:4-48
    IntervalTree(const IntervalTree& o) { *this = o; }
    IntervalTree& operator=(const IntervalTree& other);                                 // IntervalTree.cpp:50-59
    void findOverlapping(size_t start, size_t stop, intervalVector& overlapping) const; // IntervalTree.cpp:73-91

    intervalVector intervals;
    std::unique_ptr<IntervalTree> left, right;
    size_t center;
};

// ---- FMIndexWalk/SAINode.h:45-143,301-354 --------------------------------------------------
class OverlapNode {   // SAINode + SAIOverlapNode3
public:
    OverlapNode(const std::string* pQuery, OverlapNode* parent);
    OverlapNode* createChild(const std::string& label);        // SAINode.cpp:166-189
    void extend(const std::string& ext) { m_label.append(ext); }   // SAINode.cpp:73-77
    void computeInitial(const std::string& l) { m_label = l; }     // SAINode.cpp:80-84
    std::string getSuffix(size_t l) const;                         // SAINode.cpp:39-51
    std::string getFullString() const;                             // SAINode.cpp:54-60
    size_t getKmerCount() const { return m_totalKmerCount; }
    void addKmerCount(size_t c) { m_totalKmerCount += c; m_lastKmerCount = c; }   // SAINode.h:80-83

    Interval fwdInterval, rvcInterval;
    size_t lastSeedIdx;
    double numRedeemSeed;
    size_t lastOverlapLen;
    size_t totalSeeds;
    size_t currOverlapLen;
    size_t numOfErrors;
    int lastSeedIdxOffset;
    int initSeedIdx;
    size_t queryOverlapLen;
    std::pair<int, int> resultindex = std::make_pair(-1, -1);
    std::vector<double> LocalErrorRateRecord;
    std::vector<double> GlobalErrorRateRecord;

private:
    std::string m_label;
    size_t m_totalKmerCount, m_lastKmerCount;
    const std::string* m_pQuery;
    OverlapNode* m_pParent;
    std::list<std::unique_ptr<OverlapNode>> m_children3;
};

struct SAIntervalNodeResult {      // SAINode.h:174-181
    std::string thread;
    size_t SAICoverage;
    int SAIntervalSize;
    double errorRate;
};
typedef std::vector<SAIntervalNodeResult> SAIntervalNodeResultVector;

// ---- PacBio/LongReadCorrectByOverlap.h ---------------------------------------------------------
struct FMWalkResult2 {             // :21-26
    std::string mergedSeq;
    int alnScore = 0;
    double kmerFreq = 0;
};
struct FMextendParameters {        // :28-47
    IndexSet indices;
    int idmerLength = 9;
    int maxLeaves = 32;
    int minKmerLength = 13;
    size_t PBcoverage = 90;
    double ErrorRate = 0.15;
};
struct FMidx {                     // :102-151
    FMidx(const std::string& s, const Interval& f, const Interval& r)
        : SearchLetters(s), fwdInterval(f), rvcInterval(r), kmerFrequency((int)(f.size() + r.size())) {}
    FMidx(const char c, const Interval& f, const Interval& r)
        : SearchLetters(std::string(1, c)), fwdInterval(f), rvcInterval(r), kmerFrequency((int)(f.size() + r.size())) {}
    void setInterval(const Interval& f, const Interval& r)
    {
        fwdInterval = f; rvcInterval = r; kmerFrequency = (int)(f.size() + r.size());
    }
    Interval getFwdInterval() const { return fwdInterval; }
    Interval getRvcInterval() const { return rvcInterval; }
    int getKmerFrequency() const { return kmerFrequency; }
    std::string SearchLetters;
private:
    Interval fwdInterval, rvcInterval;
    int kmerFrequency;
};
typedef std::vector<FMidx> extArray;

struct leafInfo {                  // :154-217
    leafInfo(OverlapNode* leafNode, const size_t lastLeafNum);
    leafInfo(OverlapNode* currNode, const leafInfo& leaf, FMidx& extension, const size_t currLeavesNum);
    OverlapNode* leafNodePtr;
    size_t lastLeafID;
    int kmerFrequency;
    std::string tailLetter;
    size_t tailLetterCount;
};
typedef std::list<leafInfo> leafList;

// per-walk counters (not in the reference; accounting for bench/DESIGN only)
struct WalkStats {
    uint64_t steps = 0, leaf_expansions = 0, refine_calls = 0;
};

class LongReadSelfCorrectByOverlap {
public:
    LongReadSelfCorrectByOverlap(const std::string& sourceSeed, const std::string& strBetweenSrcTarget,
                                 const std::string& targetSeed, int disBetweenSrcTarget, size_t initkmersize,
                                 size_t maxOverlap, const FMextendParameters params, size_t min_SA_threshold = 3,
                                 double errorRate = 0.25, size_t repeatFreq = 256, size_t localSimilarlykmerSize = 100);
    ~LongReadSelfCorrectByOverlap();
    int extendOverlap(FMWalkResult2& FMWResult);                                        // .cpp:155-211
    WalkStats stats;

private:
    void initialRootNode(const std::string& beginningkmer);                             // :108-124
    void buildOverlapbyFMindex(IntervalTree& fwdIntervalTree, IntervalTree& rvcIntervalTree, const int& overlapSize);   // :127-152
    void extendLeaves(leafList& newLeaves);                                             // :239-278
    void attempToExtend(leafList& newLeaves, bool isSuccessToReduce);                   // :373-465
    void updateLeaves(leafList& newLeaves, extArray& extensions, leafInfo& leaf, size_t currLeavesNum);   // :468-488
    void refineSAInterval(leafList& leaves, const size_t newKmerSize);                  // :355-369
    int findTheBestPath(const SAIntervalNodeResultVector& results, FMWalkResult2& FMWResult);             // :214-236
    extArray getFMIndexExtensions(const leafInfo& currLeaf);                            // :667-784
    bool PrunedBySeedSupport(leafList& newLeaves);                                      // :491-563
    bool isInsufficientFreqs(leafList& newLeaves);                                      // :334-352
    bool isTerminated(SAIntervalNodeResultVector& results);                             // :825-878
    bool isSupportedByNewSeed(OverlapNode* currNode, size_t smallSeedIdx, size_t largeSeedIdx);           // :566-635
    bool ismatchedbykmer(Interval currFwdInterval, Interval currRvcInterval);           // :787-821
    double computeErrorRate(OverlapNode* currNode);                                     // :638-664
    size_t SelectFreqsOfrange(const size_t LowerBound, const size_t UpperBound, leafList& newLeaves);     // :281-331

    const std::string m_sourceSeed;
    const std::string m_strBetweenSrcTarget;
    const std::string m_targetSeed;
    const int m_disBetweenSrcTarget;
    const size_t m_initkmersize;
    const size_t m_minOverlap;
    const size_t m_maxOverlap;
    const RLBwt* m_pBWT;
    const RLBwt* m_pRBWT;
    const size_t m_PBcoverage;
    size_t m_min_SA_threshold;
    double m_errorRate;
    const size_t m_maxLeaves;
    const size_t m_seedSize;
    size_t m_repeatFreq;
    size_t m_localSimilarlykmerSize;
    const double m_PacBioErrorRate;
    size_t m_maxIndelSize;
    double* freqsOfKmerSize;
    size_t m_maxfreqs;
    std::string m_query;
    size_t m_maxLength;
    size_t m_minLength;
    std::vector<Interval> m_fwdTerminatedInterval;   // in rBWT
    std::vector<Interval> m_rvcTerminatedInterval;   // in BWT
    leafList m_leaves;
    OverlapNode* m_pRootNode;
    std::list<OverlapNode*> m_RootNodes;
    size_t m_currentLength;
    size_t m_currentKmerSize;
    IntervalTree m_fwdIntervalTree, m_rvcIntervalTree, m_fwdIntervalTree2, m_rvcIntervalTree2;
    size_t minTotalcount = 10000000;
    size_t totalcount = 0;
};

} // namespace lrsc_oracle
