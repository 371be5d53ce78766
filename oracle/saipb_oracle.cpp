// oracle/saipb_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  See saipb_oracle.hpp (parity unpinned).
#include "saipb_oracle.hpp"

#include <cstdlib>

#include "stdaln_oracle.hpp"

namespace lrsc_oracle {

// ---- KmerFeatures (SAIPBSelfCTree.h:42-115) ------------------------------------------------------------------
KmerFeatures::KmerFeatures(long long pos, size_t maxIntervalSize, size_t intervalSize) : m_intervalSize((long long)intervalSize)
{
    m_sumOfFreq.resize(maxIntervalSize / intervalSize + 1);
    m_sumOfPos.resize(maxIntervalSize / intervalSize + 1);
    add(pos);
}
void KmerFeatures::add(long long pos)
{
    m_totalFreq++;
    m_totalSum += pos;
    int index = (int)(pos / m_intervalSize);
    if(index < 0) index = 0;
    else if(index > (int)m_sumOfFreq.size() - 1) index = (int)m_sumOfFreq.size() - 1;
    m_sumOfFreq[(size_t)index]++;
    m_sumOfPos[(size_t)index] += pos;
}
long long KmerFeatures::getSumOfFreq(long long pos) const
{
    const int index = (int)(pos / m_intervalSize);
    // the reference indexes without a range check (.h:87-96); outside the table there is nothing to count
    if(index < 0 || index >= (int)m_sumOfFreq.size()) return 0;
    long long sumOfFreq = m_sumOfFreq[(size_t)index];
    if(index > 0) sumOfFreq += m_sumOfFreq[(size_t)index - 1];
    if(index < (int)m_sumOfFreq.size() - 1) sumOfFreq += m_sumOfFreq[(size_t)index + 1];
    return sumOfFreq;
}

// ---- SAINode / SAIntervalNode (SAINode.cpp:21-67,95-104) -------------------------------------------------------
std::string SaipbNode::getSuffix(size_t l) const
{
    const size_t n = label.size();
    if(l <= n) return label.substr(n - l, l);
    if(parent == nullptr) return label;                  // the reference asserts a parent here
    return parent->getSuffix(l - n) + label;
}
std::string SaipbNode::getFullString() const
{
    return parent == nullptr ? label : parent->getFullString() + label;
}
SaipbNode* SaipbNode::createChild(const std::string& ext)
{
    children.emplace_back(new SaipbNode());
    SaipbNode* c = children.back().get();
    c->parent = this;
    c->label = ext;
    c->addKmerCount(totalKmerCount);                     // the child inherits the parent's k-mer count
    return c;
}

// ---- the tree -----------------------------------------------------------------------------------------------
SaipbSelfCorrectTree::SaipbSelfCorrectTree(const RLBwt* bwt, const RLBwt* rbwt, const std::string& rawSeq, size_t srcmaxLength,
                                           size_t min_SA_threshold, int maxLeavesAllowed)
    : m_pBWT(bwt), m_pRBWT(rbwt), m_rawSeq(rawSeq), m_maxLength(srcmaxLength), m_min_SA_threshold(min_SA_threshold),
      m_maxLeavesAllowed((size_t)maxLeavesAllowed)
{
}

void SaipbSelfCorrectTree::insertKmerToHash(const std::string& kmer, size_t seedStrLen, size_t currentLength, size_t smallKmerSize,
                                            size_t maxLength, int expectedLength)
{
    // source to target: distance walked from the seed; target to source: the mirrored position (size_t arithmetic as written)
    const long long pos = expectedLength < 0 ? (long long)(currentLength - seedStrLen)
                                             : (long long)((size_t)expectedLength - currentLength + smallKmerSize);
    auto it = kmerHash.find(kmer);
    if(it == kmerHash.end()) kmerHash.emplace(kmer, KmerFeatures(pos, maxLength));
    else it->second.add(pos);
}

size_t SaipbSelfCorrectTree::addHashBySingleSeed(const std::string& seedStr, size_t largeKmerSize, size_t smallKmerSize, size_t maxLength,
                                                 bool skipRepeat, int expectedLength)
{
    const int64_t maxIntervalSize = 30;
    const std::string initKmer = seedStr.substr(seedStr.length() - largeKmerSize);
    const Interval fwdInterval = m_pRBWT->find_interval(reverse_str(initKmer));
    const Interval rvcInterval = m_pBWT->find_interval(reverse_complement(initKmer));
    size_t kmerFreq = 0;
    kmerFreq += fwdInterval.valid() ? (size_t)fwdInterval.size() : 0;
    kmerFreq += fwdInterval.valid() ? (size_t)rvcInterval.size() : 0;      // sic: the forward interval's validity guards both (:720)
    if(skipRepeat && kmerFreq > 128) return kmerFreq;

    // every row of the forward interval: LF-walk the reversed-read index = read the read onwards, collecting small k-mers
    for(int64_t root = fwdInterval.lower; fwdInterval.valid() && root <= fwdInterval.upper && root - fwdInterval.lower < maxIntervalSize; root++) {
        std::string cur = seedStr.substr(seedStr.length() - smallKmerSize);
        insertKmerToHash(cur, seedStr.length(), seedStr.length(), smallKmerSize, maxLength, expectedLength);
        int64_t idx = root;
        for(int64_t len = (int64_t)seedStr.length() + 1; len <= (int64_t)maxLength; len++) {
            const char b = m_pRBWT->get_char((uint64_t)idx);
            if(b == '$') break;
            cur = cur.substr(1) + b;
            insertKmerToHash(cur, seedStr.length(), (size_t)len, smallKmerSize, maxLength, expectedLength);
            idx = (int64_t)(m_pRBWT->pc(bwt_rank_of(b)) + m_pRBWT->occ(bwt_rank_of(b), idx - 1));
        }
    }
    // every row of the reverse-complement interval: LF-walk the forward index = read the other strand backwards
    for(int64_t root = rvcInterval.lower; root <= rvcInterval.upper && rvcInterval.valid() && root - rvcInterval.lower < maxIntervalSize; root++) {
        std::string cur = reverse_complement(seedStr.substr(seedStr.length() - smallKmerSize));
        insertKmerToHash(cur, seedStr.length(), seedStr.length(), smallKmerSize, maxLength, expectedLength);
        int64_t idx = root;
        for(int64_t len = (int64_t)seedStr.length() + 1; len <= (int64_t)maxLength; len++) {
            const char b = m_pBWT->get_char((uint64_t)idx);
            if(b == '$') break;
            cur = b + cur.substr(0, smallKmerSize - 1);
            insertKmerToHash(cur, seedStr.length(), (size_t)len, smallKmerSize, maxLength, expectedLength);
            idx = (int64_t)(m_pBWT->pc(bwt_rank_of(b)) + m_pBWT->occ(bwt_rank_of(b), idx - 1));
        }
    }
    return kmerFreq;
}

void SaipbSelfCorrectTree::initializeSearchTree(const std::string& src, size_t hashKmerSize)
{
    m_leaves.clear();
    m_pRootNode.reset(new SaipbNode());
    m_pRootNode->label = src;
    const std::string beginningkmer = src.substr(src.length() - hashKmerSize);
    m_pRootNode->fwd = m_pRBWT->find_interval(reverse_str(beginningkmer));
    m_pRootNode->rvc = m_pBWT->find_interval(reverse_complement(beginningkmer));
    m_leaves.push_back(m_pRootNode.get());
    m_seedLength = (int)src.length();
    m_currentLength = (int)src.length();
}

void SaipbSelfCorrectTree::initializeTerminalIntervals(const std::string& dest, size_t hashKmerSize)
{
    const std::string endingkmer = dest.substr(0, hashKmerSize);
    m_fwdTerminatedInterval = m_pRBWT->find_interval(reverse_str(endingkmer));
    m_rvcTerminatedInterval = m_pBWT->find_interval(reverse_complement(endingkmer));
}

void SaipbSelfCorrectTree::refineSAInterval(size_t newKmer)
{
    for(SaipbNode* leaf : m_leaves) {
        const std::string pkmer = leaf->getSuffix(newKmer);
        leaf->fwd = m_pRBWT->find_interval(reverse_str(pkmer));
        leaf->rvc = m_pBWT->find_interval(reverse_complement(pkmer));
    }
}

std::vector<SaipbSelfCorrectTree::Ext> SaipbSelfCorrectTree::getFMIndexRightExtensions(const SaipbNode* node, size_t IntervalSizeCutoff) const
{
    std::vector<Ext> out;
    static const char kAlphabet[] = "$ACGT";
    for(int i = 1; i < 5; ++i) {
        const char b = kAlphabet[i];
        Interval fwdProbe = node->fwd;
        if(fwdProbe.valid()) m_pRBWT->update_interval(fwdProbe, b);
        Interval rvcProbe = node->rvc;
        const char rcb = kAlphabet[5 - i];
        if(rvcProbe.valid()) m_pBWT->update_interval(rvcProbe, rcb);
        size_t bcount = 0;
        if(fwdProbe.valid()) bcount += (size_t)fwdProbe.size();
        if(rvcProbe.valid()) bcount += (size_t)rvcProbe.size();
        if(bcount >= IntervalSizeCutoff) out.push_back(Ext{b, fwdProbe, rvcProbe});
    }
    return out;
}

size_t SaipbSelfCorrectTree::hashkmerfreqs(const std::string& fwdkmer, size_t kmerposition) const
{
    const auto it1 = kmerHash.find(fwdkmer), it2 = kmerHash.find(reverse_complement(fwdkmer));
    size_t f = it1 == kmerHash.end() ? 0 : (size_t)it1->second.getSumOfFreq((long long)kmerposition);
    f += it2 == kmerHash.end() ? 0 : (size_t)it2->second.getSumOfFreq((long long)kmerposition);
    return f;
}

bool SaipbSelfCorrectTree::isExtensionValid(const std::string& fwdkmer, double& currAvgFreq, size_t& kmerFreq, size_t bcount)
{
    auto it1 = kmerHash.find(fwdkmer);
    // bubble removal, only once the frontier is wider than 8 leaves
    if(it1 != kmerHash.end() && m_leaves.size() > 8 && currAvgFreq < it1->second.getMaxAvgFreq()) return false;
    if(it1 != kmerHash.end() && currAvgFreq > it1->second.getMaxAvgFreq()) it1->second.setMaxAvgFreq(currAvgFreq);
    const auto it2 = kmerHash.find(reverse_complement(fwdkmer));
    // restricted to the local k-mer frequency: the position histogram around the distance walked so far
    kmerFreq = it1 == kmerHash.end() ? 0 : (size_t)it1->second.getSumOfFreq(m_currentLength - m_seedLength);
    kmerFreq += it2 == kmerHash.end() ? 0 : (size_t)it2->second.getSumOfFreq(m_currentLength - m_seedLength);
    return kmerFreq >= m_min_SA_threshold || (bcount >= 7 && kmerFreq >= 1);
}

void SaipbSelfCorrectTree::attempToExtendUsingHash(std::list<SaipbNode*>& newLeaves, size_t hashKmerSize, size_t minExtFreq)
{
    double maxLeafFreq = -0.1, removedMaxLeafFreq = -0.1;
    for(SaipbNode* leaf : m_leaves) {
        leaf->updated = false;
        const double currLeafFreq = (double)leaf->totalKmerCount / m_currentLength;
        if(currLeafFreq > maxLeafFreq) maxLeafFreq = currLeafFreq;
        const std::vector<Ext> extensions = getFMIndexRightExtensions(leaf, minExtFreq);
        bool isNoExtension = true;
        if(extensions.size() == 1) {
            const Ext& e = extensions.front();
            const std::string fwdkmer = leaf->getSuffix(hashKmerSize - 1) + e.b;
            double currAvgFreq = (double)leaf->totalKmerCount / (m_currentLength + 1000000);
            size_t kmerfreqs = 0;
            const size_t bcount = (size_t)(e.fwd.size() + e.rvc.size());           // raw sizes, not clamped (:1022)
            if(isExtensionValid(fwdkmer, currAvgFreq, kmerfreqs, bcount)) {
                leaf->updated = true;
                isNoExtension = false;
                leaf->label.push_back(e.b);
                leaf->fwd = e.fwd;
                leaf->rvc = e.rvc;
                leaf->addKmerCount(kmerfreqs);
                newLeaves.push_back(leaf);
            } else if(currLeafFreq > removedMaxLeafFreq)
                removedMaxLeafFreq = currLeafFreq;
        } else if(extensions.size() > 1) {
            for(const Ext& e : extensions) {
                const std::string fwdkmer = leaf->getSuffix(hashKmerSize - 1) + e.b;
                double currAvgFreq = (double)leaf->totalKmerCount / (m_currentLength + 1000000);
                size_t kmerfreqs = 0;
                const size_t bcount = (size_t)(e.fwd.size() + e.rvc.size());
                if(isExtensionValid(fwdkmer, currAvgFreq, kmerfreqs, bcount)) {
                    leaf->updated = true;
                    isNoExtension = false;
                    SaipbNode* child = leaf->createChild(std::string(1, e.b));
                    child->fwd = e.fwd;
                    child->rvc = e.rvc;
                    child->addKmerCount(kmerfreqs);
                    newLeaves.push_back(child);
                }
            }
            if(isNoExtension && currLeafFreq > removedMaxLeafFreq) removedMaxLeafFreq = currLeafFreq;
        } else if(currLeafFreq > removedMaxLeafFreq)
            removedMaxLeafFreq = currLeafFreq;
    }
    if(maxLeafFreq == removedMaxLeafFreq) m_isLargeLeaveRemoved = true;
}

bool SaipbSelfCorrectTree::isTerminated(std::vector<SaipbResult>& results)
{
    bool found = false;
    for(SaipbNode* leaf : m_leaves) {
        const Interval& f = leaf->fwd;
        const Interval& r = leaf->rvc;
        const bool isFwdTerminated = f.valid() && f.lower >= m_fwdTerminatedInterval.lower && f.upper <= m_fwdTerminatedInterval.upper;
        const bool isRvcTerminated = r.valid() && r.lower >= m_rvcTerminatedInterval.lower && r.upper <= m_rvcTerminatedInterval.upper;
        if(isFwdTerminated || isRvcTerminated) {
            results.push_back(SaipbResult{leaf->getFullString(), leaf->totalKmerCount});
            found = true;
        }
    }
    return found;
}

int SaipbSelfCorrectTree::mergeTwoSeedsUsingHash(const std::string& src, const std::string& dest, std::string& mergedseq, size_t hashKmerSize,
                                                 size_t maxLeaves, size_t minLength, size_t maxLength, size_t expectedLength)
{
    initializeSearchTree(src, hashKmerSize);
    initializeTerminalIntervals(dest, hashKmerSize);
    m_expectedLength = (int)expectedLength;
    maxUsedLeaves = 0;
    steps = 0;
    std::vector<SaipbResult> results;
    while(!m_leaves.empty() && m_leaves.size() <= maxLeaves && (size_t)m_currentLength <= maxLength) {
        ++steps;
        refineSAInterval(hashKmerSize - 1);
        std::list<SaipbNode*> newLeaves;
        attempToExtendUsingHash(newLeaves, hashKmerSize, 2);
        if(newLeaves.empty()) {
            m_min_SA_threshold--;
            attempToExtendUsingHash(newLeaves, hashKmerSize, 2);
            m_min_SA_threshold++;
        }
        if(m_leaves.size() > maxUsedLeaves) maxUsedLeaves = m_leaves.size();
        if(!newLeaves.empty()) m_currentLength++;
        m_leaves = newLeaves;
        if((size_t)m_currentLength >= minLength) isTerminated(results);
    }
    numResults = results.size();

    if(!results.empty()) {
        double maxKmerCoverage = 0, maxMatchPercent = -100;
        int minLengthDiff = 100000;
        for(const SaipbResult& res : results) {
            const std::string tmpseq = dest.length() > hashKmerSize ? res.thread + dest.substr(hashKmerSize) : res.thread;
            const int currLengthDiff = std::abs((int)tmpseq.length() - (int)expectedLength);
            const double avgCov = (double)res.SAICoverage / (tmpseq.length() + 1000000);
            const bool isLengthDiffBetter = currLengthDiff < minLengthDiff && std::abs(currLengthDiff - minLengthDiff) > 3;
            const bool isKmerCoverageBetter = std::abs(currLengthDiff - minLengthDiff) <= 3 && maxKmerCoverage < avgCov;
            if(results.size() > 1) {
                // several candidates: the one whose global alignment to the raw read matches most bases
                const int matchLen = stdaln_global_pacbio(m_rawSeq, tmpseq).matches;
                const double matchPercent = (double)matchLen / m_rawSeq.length();
                if(maxMatchPercent < matchPercent) { maxMatchPercent = matchPercent; mergedseq = tmpseq; }
            } else if(isLengthDiffBetter || isKmerCoverageBetter) {
                minLengthDiff = currLengthDiff;
                maxKmerCoverage = avgCov;
                mergedseq = tmpseq;
            }
        }
        return 1;
    }
    if(m_leaves.empty() && m_currentLength >= (int)(expectedLength - m_seedLength) / 2 + m_seedLength) return -1;   // high error
    else if((size_t)m_currentLength > maxLength) return -2;                                                     // exceed search depth
    else if(m_leaves.size() > maxLeaves) return -3;                                                             // too much repeats
    else if(m_leaves.empty() && m_currentLength < (int)(expectedLength - m_seedLength) / 2 + m_seedLength) return -4;
    return -5;
}

} // namespace lrsc_oracle
