// oracle/extend_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see extend_oracle.hpp).
#include "extend_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>

namespace lrsc_oracle {

// =======================================================================================
// IntervalTree (PacBio/IntervalTree.cpp)
// =======================================================================================
IntervalTree::IntervalTree(intervalVector& ivals, size_t depth, size_t minbucket, size_t leftextent,
                           size_t rightextent, size_t maxbucket)
    : left(nullptr), right(nullptr), center(0)
{
    (void)maxbucket;
    size_t leftp = leftextent, rightp = rightextent, centerp = 0;
    if(leftp == 0 && rightp == 0)
        // std::greater<interval> -> operator>(a,b) == a.start > b.start (IntervalTree.h:26-29).
        // libstdc++ introsort: the order it leaves equal keys in is observable downstream.
        std::sort(ivals.begin(), ivals.end(), [](const TreeInterval& a, const TreeInterval& b) { return a.start > b.start; });

    if(--depth == 0 || ivals.size() < minbucket)
        intervals = ivals;
    else {
        leftp = ivals.back().start;
        rightp = std::max_element(ivals.begin(), ivals.end(),
                                  [](const TreeInterval& a, const TreeInterval& b) { return a.stop < b.stop; })->stop;
        centerp = ivals[ivals.size() >> 1].start;
        center = centerp;

        intervalVector lefts;
        intervalVector rights;
        for(const auto& interval : ivals) {
            if(interval.stop < center)
                lefts.push_back(interval);
            else if(interval.start > center)
                rights.push_back(interval);
            else
                intervals.push_back(interval);
        }
        if(!lefts.empty()) left = std::unique_ptr<IntervalTree>(new IntervalTree(lefts, depth, minbucket, leftp, centerp));
        if(!rights.empty()) right = std::unique_ptr<IntervalTree>(new IntervalTree(rights, depth, minbucket, centerp, rightp));
    }
}

IntervalTree& IntervalTree::operator=(const IntervalTree& other)
{
    center = other.center;
    intervals = other.intervals;
    left = other.left ? std::unique_ptr<IntervalTree>(new IntervalTree(*other.left)) : nullptr;
    right = other.right ? std::unique_ptr<IntervalTree>(new IntervalTree(*other.right)) : nullptr;
    return *this;
}

void IntervalTree::findOverlapping(size_t start, size_t stop, intervalVector& overlapping) const
{
    if(!intervals.empty() && !(stop < intervals.back().start)) {
        for(const auto& interval : intervals) {
            if(interval.start <= start && interval.stop >= stop) overlapping.push_back(interval);
        }
    }
    if(left && start < center) left->findOverlapping(start, stop, overlapping);
    if(right && stop > center) right->findOverlapping(start, stop, overlapping);
}

// =======================================================================================
// OverlapNode (FMIndexWalk/SAINode.{h,cpp})
// =======================================================================================
OverlapNode::OverlapNode(const std::string* pQuery, OverlapNode* parent)
    : m_totalKmerCount(0), m_lastKmerCount(0), m_pQuery(pQuery), m_pParent(parent)
{
    lastSeedIdx = totalSeeds = lastOverlapLen = currOverlapLen = queryOverlapLen = numOfErrors = 0;
    numRedeemSeed = 0;
    lastSeedIdxOffset = 0;
    initSeedIdx = 0;
}

OverlapNode* OverlapNode::createChild(const std::string& label)
{
    OverlapNode* pAdded = new OverlapNode(m_pQuery, this);
    pAdded->extend(label);
    pAdded->lastSeedIdx = this->lastSeedIdx;
    pAdded->lastOverlapLen = this->lastOverlapLen;
    pAdded->totalSeeds = this->totalSeeds;
    pAdded->currOverlapLen = this->currOverlapLen;
    pAdded->queryOverlapLen = this->queryOverlapLen;
    pAdded->numOfErrors = this->numOfErrors;
    pAdded->lastSeedIdxOffset = this->lastSeedIdxOffset;
    pAdded->initSeedIdx = this->initSeedIdx;
    pAdded->numRedeemSeed = this->numRedeemSeed;
    pAdded->LocalErrorRateRecord = this->LocalErrorRateRecord;
    pAdded->GlobalErrorRateRecord = this->GlobalErrorRateRecord;
    pAdded->resultindex = this->resultindex;
    m_children3.push_back(std::unique_ptr<OverlapNode>(pAdded));
    return pAdded;
}

std::string OverlapNode::getSuffix(size_t l) const
{
    size_t n = m_label.size();
    if(l <= n) {
        return m_label.substr(n - l, l);
    } else {
        assert(m_pParent != NULL);
        return m_pParent->getSuffix(l - n) + m_label;
    }
}

std::string OverlapNode::getFullString() const
{
    if(m_pParent == NULL)
        return m_label;
    else
        return m_pParent->getFullString() + m_label;
}

// =======================================================================================
// leafInfo (LongReadCorrectByOverlap.h:154-217)
// =======================================================================================
leafInfo::leafInfo(OverlapNode* leafNode, const size_t lastLeafNum) : leafNodePtr(leafNode), lastLeafID(lastLeafNum)
{
    const std::string leafLabel = leafNode->getFullString();
    tailLetterCount = 0;
    for(auto reverseIdx = leafLabel.crbegin(); reverseIdx != leafLabel.crend(); ++reverseIdx) {
        std::string suffixLetter(1, (*reverseIdx));
        if(reverseIdx == leafLabel.crbegin()) tailLetter = suffixLetter;
        if(tailLetter == suffixLetter)
            tailLetterCount++;
        else
            break;
    }
    kmerFrequency = (int)((leafNode->fwdInterval).size() + (leafNode->rvcInterval).size());
}

leafInfo::leafInfo(OverlapNode* currNode, const leafInfo& leaf, FMidx& extension, const size_t currLeavesNum)
{
    const std::string& extLabel = extension.SearchLetters;
    kmerFrequency = extension.getKmerFrequency();
    currNode->fwdInterval = extension.getFwdInterval();
    currNode->rvcInterval = extension.getRvcInterval();
    currNode->addKmerCount(kmerFrequency);
    currNode->currOverlapLen++;
    currNode->queryOverlapLen++;
    leafNodePtr = currNode;
    lastLeafID = currLeavesNum;
    if(leaf.tailLetter == extLabel) {
        tailLetter = leaf.tailLetter;
        tailLetterCount = leaf.tailLetterCount + 1;
    } else {
        tailLetter = extLabel;
        tailLetterCount = 1;
    }
}

// =======================================================================================
// LongReadSelfCorrectByOverlap (PacBio/LongReadCorrectByOverlap.cpp)
// =======================================================================================
LongReadSelfCorrectByOverlap::LongReadSelfCorrectByOverlap(const std::string& sourceSeed,
                                                           const std::string& strBetweenSrcTarget,
                                                           const std::string& targetSeed, int disBetweenSrcTarget,
                                                           size_t initkmersize, size_t maxOverlap,
                                                           const FMextendParameters params, size_t min_SA_threshold,
                                                           double errorRate, size_t repeatFreq,
                                                           size_t localSimilarlykmerSize)
    : m_sourceSeed(sourceSeed),
      m_strBetweenSrcTarget(strBetweenSrcTarget),
      m_targetSeed(targetSeed),
      m_disBetweenSrcTarget(disBetweenSrcTarget),
      m_initkmersize(initkmersize),
      m_minOverlap(params.minKmerLength),
      m_maxOverlap(maxOverlap),
      m_pBWT(params.indices.bwt),
      m_pRBWT(params.indices.rbwt),
      m_PBcoverage(params.PBcoverage),
      m_min_SA_threshold(min_SA_threshold),
      m_errorRate(errorRate),
      m_maxLeaves(params.maxLeaves),
      m_seedSize(params.idmerLength),
      m_repeatFreq(repeatFreq),
      m_localSimilarlykmerSize(localSimilarlykmerSize),
      m_PacBioErrorRate(params.ErrorRate)
{
    std::string beginningkmer = m_sourceSeed.substr(m_sourceSeed.length() - m_initkmersize);

    // if distance < 100 ,use const indel size
    if(m_disBetweenSrcTarget > 100)
        m_maxIndelSize = m_disBetweenSrcTarget * 0.2;
    else
        m_maxIndelSize = 20;

    initialRootNode(beginningkmer);

    m_RootNodes.push_back(m_pRootNode);
    m_leaves.emplace_back(m_pRootNode, 1);

    // frequencies of correspond k
    freqsOfKmerSize = new double[100 + 1]{0};
    for(int i = m_minOverlap; i <= 100; i++) freqsOfKmerSize[i] = pow(1 - m_PacBioErrorRate, i) * m_PBcoverage;

    // PacBio reads are longer than real length due to insertions
    m_maxLength = (1.2 * (m_disBetweenSrcTarget + 10)) + 2 * m_initkmersize;
    m_minLength = (0.8 * (m_disBetweenSrcTarget - 20)) + 2 * m_initkmersize;

    // initialize the ending SA intervals with kmer length = m_minOverlap
    for(size_t i = 0; i <= m_targetSeed.length() - m_minOverlap; i++) {
        std::string endingkmer = m_targetSeed.substr(i, m_minOverlap);
        m_fwdTerminatedInterval.push_back(m_pRBWT->find_interval(reverse_str(endingkmer)));
        m_rvcTerminatedInterval.push_back(m_pBWT->find_interval(reverse_complement(endingkmer)));
    }
    // build overlap tree
    m_query = beginningkmer + m_strBetweenSrcTarget + m_targetSeed;
    buildOverlapbyFMindex(m_fwdIntervalTree, m_rvcIntervalTree, (int)m_seedSize);
    buildOverlapbyFMindex(m_fwdIntervalTree2, m_rvcIntervalTree2, 5);
}

LongReadSelfCorrectByOverlap::~LongReadSelfCorrectByOverlap()
{
    for(auto iter = m_RootNodes.begin(); iter != m_RootNodes.end(); ++iter) delete *iter;
    m_RootNodes.clear();
    delete[] freqsOfKmerSize;
}

void LongReadSelfCorrectByOverlap::initialRootNode(const std::string& beginningkmer)
{
    m_pRootNode = new OverlapNode(&m_sourceSeed, NULL);
    m_pRootNode->computeInitial(beginningkmer);
    m_pRootNode->fwdInterval = m_pRBWT->find_interval(reverse_str(beginningkmer));
    m_pRootNode->rvcInterval = m_pBWT->find_interval(reverse_complement(beginningkmer));
    m_pRootNode->lastOverlapLen = m_currentLength = m_pRootNode->currOverlapLen = m_pRootNode->queryOverlapLen =
        m_currentKmerSize = m_initkmersize;
    m_pRootNode->lastSeedIdx = m_pRootNode->initSeedIdx = m_initkmersize - m_seedSize;
    m_pRootNode->totalSeeds = m_initkmersize - m_seedSize + 1;
    m_pRootNode->numRedeemSeed = 0;
    m_pRootNode->LocalErrorRateRecord.push_back(0);
    m_pRootNode->GlobalErrorRateRecord.push_back(0);
    m_maxfreqs = m_pRootNode->fwdInterval.size() + m_pRootNode->rvcInterval.size();
}

void LongReadSelfCorrectByOverlap::buildOverlapbyFMindex(IntervalTree& fwdIntervalTree, IntervalTree& rvcIntervalTree,
                                                         const int& overlapSize)
{
    std::vector<TreeInterval> fwdIntervals;
    fwdIntervals.reserve(m_query.length() - overlapSize + 1);
    std::vector<TreeInterval> rvcIntervals;
    rvcIntervals.reserve(m_query.length() - overlapSize + 1);

    for(int i = 0; i <= (int)m_query.length() - (int)overlapSize; i++) {
        std::string seedStr = m_query.substr(i, overlapSize);
        Interval bi;
        bi = m_pRBWT->find_interval(reverse_str(seedStr));
        if(bi.valid()) fwdIntervals.emplace_back(bi.lower, bi.upper, i);
        bi = m_pBWT->find_interval(reverse_complement(seedStr));
        if(bi.valid()) rvcIntervals.emplace_back(bi.lower, bi.upper, i);
    }
    fwdIntervalTree = IntervalTree(fwdIntervals);
    rvcIntervalTree = IntervalTree(rvcIntervals);
}

int LongReadSelfCorrectByOverlap::extendOverlap(FMWalkResult2& FMWResult)
{
    SAIntervalNodeResultVector results;

    while(!m_leaves.empty() && m_leaves.size() <= m_maxLeaves && m_currentLength <= m_maxLength) {
        leafList newLeaves;
        extendLeaves(newLeaves);
        PrunedBySeedSupport(newLeaves);
        m_leaves.clear();
        m_leaves = newLeaves;
        if(m_currentLength >= m_minLength) isTerminated(results);
        ++stats.steps;
    }

    if(results.size() > 0) return findTheBestPath(results, FMWResult);

    if(m_leaves.empty())   // high error
        return -1;
    else if(m_currentLength > m_maxLength)   // exceed search depth
        return -2;
    else if(m_leaves.size() > m_maxLeaves)   // too much repeats
        return -3;
    else
        return -4;
}

int LongReadSelfCorrectByOverlap::findTheBestPath(const SAIntervalNodeResultVector& results, FMWalkResult2& FMWResult)
{
    double minErrorRate = 1;
    for(size_t i = 0; i < results.size(); i++) {
        const std::string& candidateSeq = results[i].thread;
        if(results[i].errorRate < minErrorRate) {
            minErrorRate = results[i].errorRate;
            FMWResult.mergedSeq = candidateSeq;
            minTotalcount = results[i].SAIntervalSize;
        }
    }
    if(FMWResult.mergedSeq.length() != 0) return 1;
    return -4;
}

void LongReadSelfCorrectByOverlap::extendLeaves(leafList& newLeaves)
{
    // resize if length too long
    if(m_currentKmerSize > m_maxOverlap) refineSAInterval(m_leaves, m_maxOverlap);

    attempToExtend(newLeaves, 1);

    if(newLeaves.empty())   // level 1 reduce size
    {
        size_t LowerBound = std::max(m_currentKmerSize - 2, m_minOverlap);
        size_t ReduceSize = SelectFreqsOfrange(LowerBound, m_currentKmerSize, m_leaves);
        bool isSuccessToReduce = m_currentKmerSize != ReduceSize;
        refineSAInterval(m_leaves, ReduceSize);

        attempToExtend(newLeaves, isSuccessToReduce);

        if(newLeaves.empty())   // level 2 reduce threshold
        {
            m_min_SA_threshold--;
            attempToExtend(newLeaves, 0);
            m_min_SA_threshold++;
        }
    }

    // extension succeed
    if(!newLeaves.empty()) {
        m_currentLength++;
        m_currentKmerSize++;
        if(isInsufficientFreqs(newLeaves))   // if frequency are low , relax it
        {
            size_t LowerBound = std::max(m_currentKmerSize - 2, m_minOverlap);
            size_t ReduceSize = SelectFreqsOfrange(LowerBound, m_currentKmerSize, newLeaves);
            refineSAInterval(newLeaves, ReduceSize);
        }
    }
}

size_t LongReadSelfCorrectByOverlap::SelectFreqsOfrange(const size_t LowerBound, const size_t UpperBound,
                                                        leafList& newLeaves)
{
    extArray maxKmerArray;
    int tempmaxfmfreqs = 0;

    for(auto& iter : newLeaves) {
        OverlapNode* leaf = iter.leafNodePtr;
        std::string maxKmer = leaf->getSuffix(UpperBound);
        std::string startkmer = maxKmer.substr(UpperBound - LowerBound);   // string of lower bound kmer size

        Interval Fwdinterval = m_pBWT->find_interval(startkmer);
        Interval Rvcinterval = m_pRBWT->find_interval(reverse_complement(reverse_str(startkmer)));

        maxKmerArray.emplace_back(maxKmer, Fwdinterval, Rvcinterval);
        FMidx& currKmer = maxKmerArray.back();
        if(currKmer.getKmerFrequency() > tempmaxfmfreqs) tempmaxfmfreqs = currKmer.getKmerFrequency();
    }

    if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound] < 5) return LowerBound;

    for(size_t i = 1; i <= UpperBound - LowerBound; i++) {
        tempmaxfmfreqs = 0;
        for(size_t j = 0; j < maxKmerArray.size(); j++) {
            std::string startkmer = maxKmerArray.at(j).SearchLetters.substr(UpperBound - LowerBound - i);
            Interval Fwdinterval = maxKmerArray.at(j).getFwdInterval();
            Interval Rvcinterval = maxKmerArray.at(j).getRvcInterval();

            char b = startkmer[0];
            char rcb = complement_base(b);
            m_pBWT->update_interval(Fwdinterval, b);
            m_pRBWT->update_interval(Rvcinterval, rcb);

            maxKmerArray.at(j).setInterval(Fwdinterval, Rvcinterval);
            if(maxKmerArray.at(j).getKmerFrequency() > tempmaxfmfreqs)
                tempmaxfmfreqs = maxKmerArray.at(j).getKmerFrequency();
        }
        if(tempmaxfmfreqs - (int)freqsOfKmerSize[LowerBound + i] < 5) return LowerBound + i;
    }
    return UpperBound;
}

bool LongReadSelfCorrectByOverlap::isInsufficientFreqs(leafList& newLeaves)
{
    size_t highfreqscount = 0;
    for(auto& iter : newLeaves) {
        int highfreqThreshold = m_PBcoverage > 60 ? (size_t)(m_PBcoverage / 60) * 3 : 3;
        if(iter.kmerFrequency > highfreqThreshold) highfreqscount++;
    }
    if(highfreqscount == 0)
        return true;
    else if(highfreqscount <= 2 && newLeaves.size() >= 5)
        return true;
    else if(highfreqscount <= 1 && newLeaves.size() >= 3)
        return true;
    return false;
}

void LongReadSelfCorrectByOverlap::refineSAInterval(leafList& leaves, const size_t newKmerSize)
{
    for(auto& iter : leaves) {
        OverlapNode* leaf = iter.leafNodePtr;
        std::string reducedKmer = leaf->getSuffix(newKmerSize);
        leaf->fwdInterval = m_pRBWT->find_interval(reverse_str(reducedKmer));
        leaf->rvcInterval = m_pBWT->find_interval(reverse_complement(reducedKmer));
        ++stats.refine_calls;
    }
    m_currentKmerSize = newKmerSize;
}

void LongReadSelfCorrectByOverlap::attempToExtend(leafList& newLeaves, bool isSuccessToReduce)
{
    double minimumErrorRate = 1;
    m_maxfreqs = 0;

    // Compute the min error rate
    for(auto& iter : m_leaves) {
        OverlapNode* leaf = iter.leafNodePtr;
        if(leaf->LocalErrorRateRecord.back() < minimumErrorRate) minimumErrorRate = leaf->LocalErrorRateRecord.back();
    }

    // Compute the errorRateDiff to trim leaves whose error rates relative to the others is high.
    leafList::iterator iter = m_leaves.begin();
    while(iter != m_leaves.end()) {
        OverlapNode* leaf = (*iter).leafNodePtr;
        double errorRateDiff = (leaf->LocalErrorRateRecord.back()) - minimumErrorRate;
        if((errorRateDiff > 0.05 && m_currentLength > m_localSimilarlykmerSize / 2) ||
           (errorRateDiff > 0.1 && m_currentLength > 15)) {
            iter = m_leaves.erase(iter);
            continue;
        }
        ++iter;
    }

    minTotalcount = 10000000;
    size_t currLeavesNum = 1;

    iter = m_leaves.begin();
    while(iter != m_leaves.end()) {
        extArray extensions;
        int count = 0;
        OverlapNode* leaf = (*iter).leafNodePtr;
        while(count < 2) {
            if(count == 1 && !(leaf->LocalErrorRateRecord.back() == minimumErrorRate && m_leaves.size() > 1)) break;

            extensions = getFMIndexExtensions(*iter);

            if(extensions.size() > 0) {
                updateLeaves(newLeaves, extensions, *iter, currLeavesNum);
                break;
            }
            isSuccessToReduce = false;
            m_min_SA_threshold--;
            count++;
        }
        m_min_SA_threshold += count;

        if(minTotalcount >= totalcount) {
            minTotalcount = totalcount;
        }
        ++iter;
        ++currLeavesNum;
    }
    (void)isSuccessToReduce;
}

void LongReadSelfCorrectByOverlap::updateLeaves(leafList& newLeaves, extArray& extensions, leafInfo& leaf,
                                                size_t currLeavesNum)
{
    OverlapNode* pNode = leaf.leafNodePtr;
    if(extensions.size() == 1) {
        // Single extension, do not branch
        pNode->extend(extensions.front().SearchLetters);
        newLeaves.emplace_back(pNode, leaf, extensions.front(), currLeavesNum);
    } else if(extensions.size() > 1) {
        // Branch
        for(size_t i = 0; i < extensions.size(); ++i) {
            OverlapNode* pChildNode = pNode->createChild(extensions[i].SearchLetters);
            // inherit accumulated kmerCount from parent
            pChildNode->addKmerCount(pNode->getKmerCount());
            newLeaves.emplace_back(pChildNode, leaf, extensions[i], currLeavesNum);
        }
    }
}

bool LongReadSelfCorrectByOverlap::PrunedBySeedSupport(leafList& newLeaves)
{
    size_t currSeedIdx = m_currentLength - m_seedSize;
    size_t indelOffset = m_seedSize + m_maxIndelSize;

    // Compute the range of small and large indices for tolerating m_maxIndelSize
    size_t smallSeedIdx = currSeedIdx <= indelOffset ? 0 : currSeedIdx - indelOffset;
    size_t largeSeedIdx = (currSeedIdx + indelOffset) >= (m_query.length() - m_seedSize) ? (m_query.length() - m_seedSize)
                                                                                           : currSeedIdx + indelOffset;

    leafList::iterator iter = newLeaves.begin();
    while(iter != newLeaves.end()) {
        bool isNewSeedFound = false;
        OverlapNode* leaf = (*iter).leafNodePtr;

        if(m_currentLength - leaf->lastOverlapLen > m_seedSize || m_currentLength - leaf->lastOverlapLen <= 1) {
            size_t preSeedIdx = leaf->lastSeedIdx;
            // search for matched new seeds
            isNewSeedFound = isSupportedByNewSeed(leaf, smallSeedIdx, largeSeedIdx);

            // lastSeedIdxOffset records the offset between lastSeedIdx and currSeedIdx when first match is found
            if(isNewSeedFound) {
                if(currSeedIdx + leaf->lastSeedIdxOffset - preSeedIdx > m_seedSize)
                    leaf->numRedeemSeed += (m_seedSize - 1) * m_PacBioErrorRate;

                leaf->lastSeedIdxOffset = (int)leaf->lastSeedIdx - (int)currSeedIdx;
            } else {
                if((currSeedIdx + leaf->lastSeedIdxOffset - leaf->lastSeedIdx) % m_seedSize == 1)
                    leaf->numOfErrors++;
                else if((currSeedIdx + leaf->lastSeedIdxOffset - leaf->lastSeedIdx) > m_seedSize - 1)
                    leaf->numRedeemSeed += 1 - m_PacBioErrorRate;
            }
        } else
            leaf->numRedeemSeed += 1 - m_PacBioErrorRate;

        double currErrorRate = computeErrorRate(leaf);

        // This is the 2nd filter less reliable than the 1st one
        if(currErrorRate > m_errorRate) {
            iter = newLeaves.erase(iter);
            continue;
        }
        iter++;
    }
    return true;
}

bool LongReadSelfCorrectByOverlap::isSupportedByNewSeed(OverlapNode* currNode, size_t smallSeedIdx, size_t largeSeedIdx)
{
    // If there is mismatch/indel, jump to the next m_seedSize/m_seedDist, and 1 otherwise.
    size_t seedIdxOffset = currNode->lastOverlapLen < m_currentLength - m_seedSize ? m_seedSize
                                                                                   : m_currentLength - currNode->lastOverlapLen;

    // search for new seed starting from last matched seed or smallSeedIdx
    size_t startSeedIdx = std::max(smallSeedIdx, currNode->lastSeedIdx + seedIdxOffset);

    bool isNewSeedFound = false;
    Interval currFwdInterval = currNode->fwdInterval;
    Interval currRvcInterval = currNode->rvcInterval;

    std::vector<TreeInterval> resultsFwd, resultsRvc;
    if(currFwdInterval.valid()) m_fwdIntervalTree.findOverlapping(currFwdInterval.lower, currFwdInterval.upper, resultsFwd);
    if(currRvcInterval.valid()) m_rvcIntervalTree.findOverlapping(currRvcInterval.lower, currRvcInterval.upper, resultsRvc);
    int minIdxDiff = 10000;
    size_t currSeedIdx = m_currentLength - m_seedSize;
    for(size_t i = 0; i < resultsFwd.size() || i < resultsRvc.size(); i++) {
        if(currFwdInterval.valid() && i < resultsFwd.size() && resultsFwd.at(i).value >= startSeedIdx &&
           resultsFwd.at(i).value <= largeSeedIdx) {
            if(std::abs((int)resultsFwd.at(i).value - (int)currSeedIdx) < minIdxDiff) {
                currNode->lastSeedIdx = resultsFwd.at(i).value;
                // query overlap may shift due to indels
                currNode->queryOverlapLen = resultsFwd.at(i).value + m_seedSize;
                minIdxDiff = std::abs((int)resultsFwd.at(i).value - (int)currSeedIdx);
            }
            currNode->lastOverlapLen = m_currentLength;
            currNode->currOverlapLen = m_currentLength;
            isNewSeedFound = true;
        } else if(currRvcInterval.valid() && i < resultsRvc.size() && resultsRvc.at(i).value >= startSeedIdx &&
                  resultsRvc.at(i).value <= largeSeedIdx) {
            if(std::abs((int)currSeedIdx - (int)resultsRvc.at(i).value) < minIdxDiff) {
                currNode->lastSeedIdx = resultsRvc.at(i).value;
                currNode->queryOverlapLen = resultsRvc.at(i).value + m_seedSize;
                minIdxDiff = std::abs((int)resultsRvc.at(i).value - (int)currSeedIdx);
            }
            currNode->lastOverlapLen = m_currentLength;
            currNode->currOverlapLen = m_currentLength;
            isNewSeedFound = true;
        }
    }

    if(isNewSeedFound) currNode->totalSeeds++;
    return isNewSeedFound;
}

double LongReadSelfCorrectByOverlap::computeErrorRate(OverlapNode* currNode)
{
    // Compute accuracy via matched length in both query and subject
    double matchedLen = (double)currNode->totalSeeds + m_seedSize - 1;
    matchedLen += currNode->numRedeemSeed;
    double totalLen = (double)currNode->currOverlapLen;
    double unmatchedLen = totalLen - matchedLen;
    double currErrorRate = unmatchedLen / totalLen;
    currNode->GlobalErrorRateRecord.push_back(currErrorRate);

    if(currNode->GlobalErrorRateRecord.size() >= m_localSimilarlykmerSize) {
        size_t totalsize = currNode->GlobalErrorRateRecord.size();
        currErrorRate = (currErrorRate * totalLen - currNode->GlobalErrorRateRecord.at(totalsize - m_localSimilarlykmerSize) *
                                                        (totalLen - m_localSimilarlykmerSize)) /
                        m_localSimilarlykmerSize;
    }
    currNode->LocalErrorRateRecord.push_back(currErrorRate);
    return currErrorRate;
}

extArray LongReadSelfCorrectByOverlap::getFMIndexExtensions(const leafInfo& currLeaf)
{
    OverlapNode* leaf = currLeaf.leafNodePtr;
    extArray output;
    output.reserve(4);
    extArray totalExt;
    totalExt.reserve(4);
    ++stats.leaf_expansions;

    size_t IntervalSizeCutoff = m_min_SA_threshold;   // min freq at fwd and rvc bwt, >=3 is equal to >=2 kmer freq

    totalcount = 0;
    int maxfreqsofleave = 0;

    for(int i = 1; i < 5; ++i)   // i=A,C,G,T
    {
        // update forward Interval using extension b
        char b = bwt_char_of(i);
        Interval fwdProbe = leaf->fwdInterval;
        if(fwdProbe.valid()) m_pRBWT->update_interval(fwdProbe, b);

        // update reverse complement Interval using extension rcb
        char rcb = bwt_char_of(5 - i);
        Interval rvcProbe = leaf->rvcInterval;
        if(rvcProbe.valid()) m_pBWT->update_interval(rvcProbe, rcb);

        FMidx currExt = FMidx(b, fwdProbe, rvcProbe);
        totalcount += currExt.getKmerFrequency();
        if(currExt.getKmerFrequency() > maxfreqsofleave) maxfreqsofleave = currExt.getKmerFrequency();
        totalExt.push_back(currExt);
    }

    m_maxfreqs = std::max(m_maxfreqs, totalcount);

    for(int i = 1; i < 5; ++i) {
        size_t kmerFreq = totalExt.at(i - 1).getKmerFrequency();
        Interval fwdInterval = totalExt.at(i - 1).getFwdInterval();
        Interval rvcInterval = totalExt.at(i - 1).getRvcInterval();

        const double kmerRatioNotPass = 2;
        double kmerRatioCutoff = 0;
        double kmerRatio = (double)kmerFreq / (double)maxfreqsofleave;

        char b = bwt_char_of(i);

        bool isHomopolymer = (currLeaf.tailLetterCount >= 3);
        bool isMatchedBy5mer = ismatchedbykmer(fwdInterval, rvcInterval);

        bool isFreqPass = kmerFreq >= IntervalSizeCutoff;
        bool isLowCoverage = totalcount >= IntervalSizeCutoff + 2;
        bool isRepeat = maxfreqsofleave > 100;
        bool isHighlyRepeat = maxfreqsofleave > 150;
        bool isLowlyRepeat = maxfreqsofleave > 50;

        if(isMatchedBy5mer && isHighlyRepeat)
            kmerRatioCutoff = 0.125;
        else if(isMatchedBy5mer && isLowlyRepeat)
            kmerRatioCutoff = 0.2;
        else if(isFreqPass)
            kmerRatioCutoff = 0.25;
        else if(isLowCoverage)
            kmerRatioCutoff = 0.6;
        else
            kmerRatioCutoff = kmerRatioNotPass;

        if(isHomopolymer && isRepeat)
            kmerRatioCutoff = std::max(kmerRatioCutoff, 0.3);
        else if(isHomopolymer)
            kmerRatioCutoff = std::max(kmerRatioCutoff, 0.6);

        if(kmerRatio >= kmerRatioCutoff) {
            output.emplace_back(b, fwdInterval, rvcInterval);
        }
    }
    return output;
}

bool LongReadSelfCorrectByOverlap::ismatchedbykmer(Interval currFwdInterval, Interval currRvcInterval)
{
    bool match = false;
    std::vector<TreeInterval> resultsFwd, resultsRvc;
    if(currFwdInterval.valid()) m_fwdIntervalTree2.findOverlapping(currFwdInterval.lower, currFwdInterval.upper, resultsFwd);
    if(currRvcInterval.valid()) m_rvcIntervalTree2.findOverlapping(currRvcInterval.lower, currRvcInterval.upper, resultsRvc);
    size_t startSeedIdx = std::max((int)m_currentLength - (int)m_maxIndelSize, 0);
    size_t largeSeedIdx = m_currentLength + m_maxIndelSize;

    for(size_t i = 0; i < resultsFwd.size() || i < resultsRvc.size(); i++) {
        if(currFwdInterval.valid() && i < resultsFwd.size() && resultsFwd.at(i).value >= startSeedIdx &&
           resultsFwd.at(i).value <= largeSeedIdx) {
            match = true;
            break;
        } else if(currRvcInterval.valid() && i < resultsRvc.size() && resultsRvc.at(i).value >= startSeedIdx &&
                  resultsRvc.at(i).value <= largeSeedIdx) {
            match = true;
            break;
        }
    }
    return match;
}

bool LongReadSelfCorrectByOverlap::isTerminated(SAIntervalNodeResultVector& results)
{
    bool found = false;
    for(leafList::iterator iter = m_leaves.begin(); iter != m_leaves.end(); ++iter) {
        OverlapNode* leaf = (*iter).leafNodePtr;
        Interval currfwd = leaf->fwdInterval;
        Interval currrvc = leaf->rvcInterval;

        assert(currfwd.valid() || currrvc.valid());

        bool isFwdTerminated = false;
        bool isRvcTerminated = false;
        for(size_t i = std::max(leaf->resultindex.second, 0); i <= m_targetSeed.length() - (int)m_minOverlap; i++) {
            isFwdTerminated = currfwd.valid() && currfwd.lower >= m_fwdTerminatedInterval.at(i).lower &&
                              currfwd.upper <= m_fwdTerminatedInterval.at(i).upper;
            isRvcTerminated = currrvc.valid() && currrvc.lower >= m_rvcTerminatedInterval.at(i).lower &&
                              currrvc.upper <= m_rvcTerminatedInterval.at(i).upper;

            if(isFwdTerminated || isRvcTerminated) {
                std::string STNodeStr = leaf->getFullString();
                if(m_targetSeed.length() > m_minOverlap) STNodeStr += m_targetSeed.substr(i + m_minOverlap);

                SAIntervalNodeResult STresult;
                STresult.thread = STNodeStr;
                STresult.SAICoverage = leaf->getKmerCount();
                STresult.errorRate = leaf->GlobalErrorRateRecord.back();
                STresult.SAIntervalSize = (int)(currfwd.upper - currfwd.lower + 1);

                if(leaf->resultindex.first == -1) {
                    results.push_back(STresult);
                    leaf->resultindex = std::make_pair((int)results.size(), (int)i);
                } else {
                    results.at(leaf->resultindex.first - 1) = STresult;
                    leaf->resultindex = std::make_pair(leaf->resultindex.first, (int)i);
                }
                found = true;
            }
        }
    }
    return found;
}

} // namespace lrsc_oracle
